# round 6, final build: every profiles/r06_* file again (tools/gpurun/r6_profiles.sh), then the longer fuzz runs with other seeds
cd $GRAFT_REPO_ROOT
bash tools/gpurun/r6_profiles.sh
timeout 1500 python tests/fuzz_parity.py ${R6_FUZZ:-300} 40000 2>&1 | tail -1
timeout 1200 python tests/fuzz_sharded.py ${R6_FUZZ_SHARDED:-120} 50000 2>&1 | tail -1
