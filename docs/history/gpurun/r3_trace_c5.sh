# kernel trace of a C5 step: gpurun_out/step_trace_c5.txt
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/gaps5 -- python3 $ROOT/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $ROOT/gpurun_out/gaps5_bench.json 2> $ROOT/gpurun_out/gaps5.log
f=$(ls $ROOT/gpurun_out/gaps5/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/trace_gaps.py $f ALL > $ROOT/gpurun_out/step_trace_c5.txt
rm -rf $ROOT/gpurun_out/gaps5
