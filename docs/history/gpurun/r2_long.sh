# long reads: parity of the chain's second kernel + the long-read probe
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_reads or full_path_small or degenerate" > gpurun_out/r2_long_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r2_long_pytest.log | tail -3
timeout 600 python tools/long_read_probe.py 2>&1 | grep "^x"
