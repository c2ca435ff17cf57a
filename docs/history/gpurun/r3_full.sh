# the whole GPU suite as the driver runs it, log in gpurun_out/full.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/full.log 2>&1
tail -5 gpurun_out/full.log
