# round 6: the counting pass in windows of groups (more than 4.9 M reads): parity, fuzzers; bucketing's time unchanged at C3
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_rows.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 900 python tests/fuzz_parity.py 80 2>&1 | tail -1
timeout 900 python tests/fuzz_sharded.py 30 2>&1 | tail -1
timeout 600 python tools/bucket_shift_ab.py c3 0,12 3 2>&1 | tail -2
