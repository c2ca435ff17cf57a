cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_cli.py -x -q -k "sensitive or several_ranks" > gpurun_out/sens_pytest.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/sens_pytest.log | tail -15
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -k "sensitive" 2>&1 | tail -4
RALA_HIP_TRACE=1 python tools/sens_bench.py c3 2>&1 | grep -v "^\[trace\] tail\|containment round" | tail -30
