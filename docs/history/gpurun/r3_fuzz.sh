# differential fuzzing on the GPU box: tests/fuzz_parity.py <cases> <first seed>, log in gpurun_out/fuzz.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tests/fuzz_parity.py ${1:-150} ${2:-31000} > gpurun_out/fuzz.log 2>&1
echo "exit $?" >> gpurun_out/fuzz.log
tail -4 gpurun_out/fuzz.log
