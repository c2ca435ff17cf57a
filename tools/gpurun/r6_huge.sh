# round 6: more than 2^30 overlaps in one context (row offsets in bound pairs) - tools/huge_run.py; first the tool on a small set
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_offsets or partition_sizes" 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python tools/huge_run.py 60000 6000000 6 2>&1 | grep "^\[huge\]\|Error\|error" | tail -6
timeout ${R6_HUGE_TIMEOUT:-2400} python tools/huge_run.py ${R6_HUGE_READS:-4500000} ${R6_HUGE_GENOME:-176000000} 6 gpurun_out/r06_beyond_2_30_overlaps.json 2>&1 | grep -v "^{" | tail -14
