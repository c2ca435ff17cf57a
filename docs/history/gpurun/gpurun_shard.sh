ROOT=$GRAFT_REPO_ROOT
cd $ROOT && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3
RALA_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/shard.log | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('sharded(world=1):', d['value'], d['ms_per_step'], d['stage_ms'])"
tail -5 gpurun_out/shard.log
