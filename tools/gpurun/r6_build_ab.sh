# round 6: two BUILDS of the library in alternating processes on one box (tools/ab/*.so, built here from the named commits): the
# pile chain's and the bucketing's time of the product path, tools/pile_ab.py.  The other build: `git worktree add /tmp/old 6eea2dc`, build
# it there (rala_amd.build.build_hip()), copy its librala_hip.so to tools/ab/librala_hip_6eea2dc.so (not kept in the tree).
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  for lib in tools/ab/librala_hip_6eea2dc.so ""; do
    if [ -n "$lib" ]; then export RALA_HIP_LIB_AB=$GRAFT_REPO_ROOT/$lib; name=6eea2dc; else unset RALA_HIP_LIB_AB; name=head; fi
    timeout 300 python tools/pile_ab.py c3 0 3 4 2>&1 | grep "variant 0\|bucketing" | tr '\n' ' ' | sed "s/^/$name $i: /"; echo
  done
done
