# round 5: the whole GPU suite, the fuzzers, then every profiles/r05_* file
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python tests/fuzz_parity.py 150 2>&1 | tail -2
timeout 900 python tests/fuzz_sharded.py 60 2>&1 | tail -2
bash tools/gpurun/r5_profiles.sh
