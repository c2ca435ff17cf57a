cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out tools/_bin
export TMPDIR=/tmp
RALA_IO_TRACE=1 python tools/e2e_bench.py c3 > gpurun_out/r2_e2e_c3.json 2> gpurun_out/r2_e2e_c3.log
grep "\[io\]\|\[e2e\]" gpurun_out/r2_e2e_c3.log | tail -6; cat gpurun_out/r2_e2e_c3.json
