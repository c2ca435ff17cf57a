# what the round-end driver runs, in its order
T0=$(date +%s); lap() { T1=$(date +%s); echo "[$1: $((T1 - T0)) s]"; T0=$T1; }
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/driver_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/driver_pytest.log | tail -3
lap "pytest gpu"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
lap "smoke"
python bench.py > gpurun_out/driver_bench.json 2> gpurun_out/driver_bench.log; tail -1 gpurun_out/driver_bench.log; cut -c1-400 gpurun_out/driver_bench.json
lap "bench default"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 2>gpurun_out/driver_tr.log | tail -1 | cut -c1-300; tail -1 gpurun_out/driver_tr.log
lap "bench torchrun n=1"
