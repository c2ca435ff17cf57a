cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RALA_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload c2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -4
