# round 6: does a process' slow state need kernels running beside the big one?  Processes alternating between the product's
# streams (main + aux + side) and everything on one stream (option use_side_stream = 0); and with eight hardware queues
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  timeout 300 python tools/pile_ab.py c3 0 3 4 2>&1 | grep "variant 0" | sed "s/^/streams $i: /"
  RALA_AB_OPTIONS=use_side_stream=0 timeout 300 python tools/pile_ab.py c3 0 3 4 2>&1 | grep "variant 0" | sed "s/^/one stream $i: /"
  GPU_MAX_HW_QUEUES=8 timeout 300 python tools/pile_ab.py c3 0 3 4 2>&1 | grep "variant 0" | sed "s/^/8 queues $i: /"
done
