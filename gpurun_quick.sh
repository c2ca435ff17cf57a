cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -4
python tools/phase_probe.py c2 23,24
python tools/phase_probe.py c3 99
