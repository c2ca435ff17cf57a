# round 6: inside ONE process, do contexts whose buffers land elsewhere differ as much as processes do?  (tools/pile_alloc_probe.py,
# the product kernel only; blocks of 7 / 21 GB stay allocated between the contexts)
cd $GRAFT_REPO_ROOT
for k in 1 2 3 4; do echo "process $k:"; timeout 300 python tools/pile_alloc_probe.py c3 0 5 3 2>&1 | grep "^context"; done
