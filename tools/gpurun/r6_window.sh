cd $GRAFT_REPO_ROOT
timeout 300 python tools/window_fill_probe.py 32 2>&1 | tail -6
