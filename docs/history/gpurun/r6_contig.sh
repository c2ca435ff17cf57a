# round 6: is the box's "slow state" the pile buffer's physical placement?  The same step in processes that alternate between the
# default allocation and a physically contiguous pile buffer (hipDeviceMallocContiguous); and the translation counters there are
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -o "[A-Z0-9_]*\(UTCL\|TLB\|XNACK\|TRANSLAT\)[A-Za-z0-9_]*" | sort -u | head -40
for i in 1 2 3 4 5; do
  for mode in default contiguous; do
    if [ $mode = contiguous ]; then export RALA_HIP_PILE_CONTIGUOUS=1; else unset RALA_HIP_PILE_CONTIGUOUS; fi
    python tools/pile_ab.py c3 0 3 4 2>&1 | grep "variant 0" | sed "s/^/$mode $i: /"
  done
done
