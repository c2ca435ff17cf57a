# round 5: the step trace of the sharded runner with a world of one (RCCL), C3
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05t
mkdir -p $OUT
export RALA_FORCE_SHARDED=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_sh -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/tr_sh.log
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/tr_sh/*/*kernel_trace.csv | head -1) ALL > $OUT/c3_sharded_step_trace.txt
rm -rf $OUT/tr_sh
sed -n 1,140p $OUT/c3_sharded_step_trace.txt | cut -c1-140
