# differential fuzzing of the sharded runner on the GPU box: tests/fuzz_sharded.py <cases> <first seed>, log in gpurun_out/fuzz_sharded.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tests/fuzz_sharded.py ${1:-100} ${2:-7000} > gpurun_out/fuzz_sharded.log 2>&1
echo "exit $?" >> gpurun_out/fuzz_sharded.log
grep -c ": ok\|all filtered" gpurun_out/fuzz_sharded.log
grep "MISMATCH" gpurun_out/fuzz_sharded.log | head -10
tail -3 gpurun_out/fuzz_sharded.log
