cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out tools/_bin
export TMPDIR=/tmp
RALA_IO_TRACE=1 python tools/e2e_bench.py c3 > gpurun_out/r2_e2e_c3.json 2> gpurun_out/r2_e2e_c3.log
grep "\[io\] 16\|\[e2e\]" gpurun_out/r2_e2e_c3.log | tail -4; cat gpurun_out/r2_e2e_c3.json
if [ -n "$1" ]; then
RALA_IO_NO_PIN=1 RALA_IO_TRACE=1 python tools/e2e_bench.py c3 > gpurun_out/r2_e2e_c3_nopin.json 2> gpurun_out/r2_e2e_c3_nopin.log
grep "\[io\] 16\|\[e2e\]" gpurun_out/r2_e2e_c3_nopin.log | tail -4; cat gpurun_out/r2_e2e_c3_nopin.json
fi
