# round 6: the reserved range on a chunk's boundary (1 GB) or wherever the runtime puts it: placements per process as before
cd $GRAFT_REPO_ROOT
run() {
  echo "== $1"
  for k in 1 2; do RALA_HIP_TRACE_BUFFERS=1 timeout 400 python tools/pile_alloc_probe.py c3 0,262144 4 3 2>&1 | awk '/piles 0x/ { match($0, /piles 0x[0-9a-f]+/); at = substr($0, RSTART + 6, RLENGTH - 6) } /^context/ { sub(/\(free before: [0-9.]* GB, /, ""); sub(/\)/, ""); print $0 "  rows at " at }' | cut -c1-170; done
}
unset RALA_HIP_RANGE_ALIGNED
run "range where the runtime puts it"
export RALA_HIP_RANGE_ALIGNED=1
run "range on a 1 GB boundary"
