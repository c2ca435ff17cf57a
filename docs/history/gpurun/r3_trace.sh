# the whole C3 step in time order (kernel trace of a short bench run): gpurun_out/step_trace.txt
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/gaps -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > $ROOT/gpurun_out/gaps_bench.json 2> $ROOT/gpurun_out/gaps.log
f=$(ls $ROOT/gpurun_out/gaps/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/trace_gaps.py $f ALL > $ROOT/gpurun_out/step_trace.txt
head -30 $ROOT/gpurun_out/step_trace.txt
rm -rf $ROOT/gpurun_out/gaps
