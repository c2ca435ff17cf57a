# round 6: the first-level partition size chosen at run time (at most 256 partitions): parity with every size, both fuzzers, and the
# bucketing's time at C5 (4 096 against 16 384 reads per partition) and C3 (unchanged: 245 partitions of 4 096)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_wrap.py tests/test_gpu_rows.py tests/test_gpu_sharded.py tests/test_gpu_unbounded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 900 python tests/fuzz_parity.py 60 2>&1 | tail -1
timeout 900 python tests/fuzz_sharded.py 30 2>&1 | tail -1
timeout 900 python tools/bucket_shift_ab.py c5 12,0,13 4 2>&1 | tail -4
timeout 600 python tools/bucket_shift_ab.py c3 12,0,13,14 4 2>&1 | tail -5
