cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q -k "windows" 2>&1 | tail -60
