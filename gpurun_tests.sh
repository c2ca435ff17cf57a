cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
bash gpurun_bench.sh
