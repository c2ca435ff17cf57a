cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/phase_probe.py c2 22,25,26,27
bash tools/gpurun/gpurun_bench.sh
