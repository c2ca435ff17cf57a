# round 5: a longer run of both fuzzers on the final build (other seeds than the suites')
cd $GRAFT_REPO_ROOT
timeout 2400 python tests/fuzz_parity.py 300 20000 2>&1 | tail -2
timeout 1800 python tests/fuzz_sharded.py 120 30000 2>&1 | tail -2
