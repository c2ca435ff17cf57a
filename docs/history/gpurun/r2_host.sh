cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_cli.py tests/test_gpu_host_api.py tests/test_gpu_sharded.py -x -q > gpurun_out/host_pytest.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/host_pytest.log | tail -30
RALA_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload c3 --steps 5 --warmup 1 --no-cpu-baseline 2>gpurun_out/shard_bench.log | tail -1 > gpurun_out/shard_bench_world1.json
python -c "import json; d=json.load(open('gpurun_out/shard_bench_world1.json')); print('sharded(world=1):', d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()})" || tail -20 gpurun_out/shard_bench.log
