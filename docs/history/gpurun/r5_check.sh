# round 5: the whole GPU suite and both fuzzers (after a kernel change)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 900 python tests/fuzz_parity.py 100 40000 2>&1 | tail -1
timeout 900 python tests/fuzz_sharded.py 40 50000 2>&1 | tail -1
