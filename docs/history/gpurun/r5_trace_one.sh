# round 5: the step trace of one workload ($1, default c3) with the environment as given -> gpurun_out/r05t/
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05t
mkdir -p $OUT
W=${1:-c3}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$W -- python3 $ROOT/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/tr_$W.log
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/tr_$W/*/*kernel_trace.csv | head -1) ALL > $OUT/${W}_step_trace.txt
rm -rf $OUT/tr_$W
head -60 $OUT/${W}_step_trace.txt | cut -c1-150
