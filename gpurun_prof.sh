ROOT=$GRAFT_REPO_ROOT
cd $ROOT && python bench.py --steps 5 --warmup 1 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log; cat gpurun_out/bench_default.json | cut -c1-2200
RALA_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload c3 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('sharded(world=1):', d['value'], d['ms_per_step'], d['stage_ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_c3b -- python3 $ROOT/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_c3b_bench.json 2> $ROOT/gpurun_out/prof_c3b.log
rm -f $ROOT/gpurun_out/prof_c3b/*/*kernel_trace.csv
