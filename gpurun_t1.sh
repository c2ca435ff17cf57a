cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_golden.py -x -q -m gpu 2>&1 | grep -E "Error|error|assert|mismatch|golden" | head -20
