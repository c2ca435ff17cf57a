# round 6: the pile buffer physically contiguous (a deterministic placement): which mapping of rows to workgroups suits it?
# variants: 0 XCD ranges, 8192 reads as launched, 65536 XCD ranges + nt stores, 73728 as launched + nt
cd $GRAFT_REPO_ROOT
export R6_FLAGS="'-DRALA_PILE_AB_CASES=X(8192) X(65536) X(73728)'"
echo "== contiguous"
RALA_HIP_PILE_CONTIGUOUS=1 R6_PROCS=2 R6_ROUNDS=3 R6_STEPS=3 bash tools/gpurun/r6_ab_inproc.sh 0,8192,65536,73728
echo "== default"
R6_PROCS=2 R6_ROUNDS=3 R6_STEPS=3 bash tools/gpurun/r6_ab_inproc.sh 0,8192,65536,73728
