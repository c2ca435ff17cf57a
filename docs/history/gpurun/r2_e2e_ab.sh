# end to end from PAF, alternating a reader variant (environment variable $1 = 1) inside one process
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RALA_E2E_AB=$1 RALA_IO_TRACE=1 python tools/e2e_bench.py c3 > gpurun_out/r2_e2e_ab.json 2> gpurun_out/r2_e2e_ab.log
grep "\[e2e\]\|\[io\] 16" gpurun_out/r2_e2e_ab.log | tail -20
