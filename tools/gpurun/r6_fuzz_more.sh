# round 6: a longer run of both fuzzers on the final build (other seeds than the suites'), every case with the every-row check
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_wrap.py -m gpu -x -q 2>&1 | tail -2
timeout 2400 python tests/fuzz_parity.py ${R6_FUZZ:-300} 40000 2>&1 | tail -2
timeout 1800 python tests/fuzz_sharded.py ${R6_FUZZ_SHARDED:-120} 50000 2>&1 | tail -2
