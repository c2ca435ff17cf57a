cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_sharded.py -x -q > gpurun_out/shard_pytest.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/shard_pytest.log | tail -40
