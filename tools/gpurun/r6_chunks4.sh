# round 6: the rows in mapped chunks as the product's default: what the allocation costs, the parity suites, both fuzzers
cd $GRAFT_REPO_ROOT
timeout 300 python tools/alloc_time.py c3 2>&1 | tail -16
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_gpu_sharded.py tests/test_gpu_edges.py tests/test_gpu_wrap.py tests/test_gpu_unbounded.py tests/test_gpu_host_api.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 900 python tests/fuzz_parity.py 60 2>&1 | tail -1
timeout 900 python tests/fuzz_sharded.py 30 2>&1 | tail -1
