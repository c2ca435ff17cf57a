# idle gaps inside a C3 step (host round trips, launch latency): kernel trace of a short bench run
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/gaps -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/gaps_bench.json 2> $ROOT/gpurun_out/gaps.log
f=$(ls $ROOT/gpurun_out/gaps/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/trace_gaps.py $f $1 | tee $ROOT/gpurun_out/gaps_summary.txt
rm -rf $ROOT/gpurun_out/gaps
