set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --workload c2 --steps 3 --warmup 1 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.log; tail -5 gpurun_out/bench_c2.log; cat gpurun_out/bench_c2.json
python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.log; tail -5 gpurun_out/bench_c3.log; cat gpurun_out/bench_c3.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_c2 -name "*stats*" | head; 
