cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python tools/phase_probe.py c2 21,22,23,24,25,26,27,28
bash tools/gpurun/gpurun_bench.sh 2>&1 | grep -v trace
