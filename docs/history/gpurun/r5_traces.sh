# round 5: the step traces of C3, c3s, c5s (kernel trace -> tools/trace_gaps.py)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in c3 c3s c5s; do
  steps=6; [ $w = c5s ] && steps=3
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$w -- python3 $ROOT/bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/tr_$w.log
  python3 $ROOT/tools/trace_gaps.py $(ls $OUT/tr_$w/*/*kernel_trace.csv | head -1) ALL > $OUT/r05_${w}_step_trace.txt
  rm -rf $OUT/tr_$w
  head -2 $OUT/r05_${w}_step_trace.txt
done
