# round 6: the whole GPU suite, the fuzzers, then every profiles/r06_* file (= tools/gpurun/r6_profiles.sh)
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python tests/fuzz_parity.py ${R6_FUZZ:-150} 2>&1 | tail -2
timeout 900 python tests/fuzz_sharded.py ${R6_FUZZ_SHARDED:-60} 2>&1 | tail -2
bash tools/gpurun/r6_profiles.sh
