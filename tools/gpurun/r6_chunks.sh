# round 6: the rows' buffer as physical chunks mapped side by side (hipMemCreate / hipMemMap) instead of one hipMalloc: does the
# size of the chunks, or their order, decide where the pile kernel lands between 3.8 and 4.3 ms?  Five contexts per process
# (tools/pile_alloc_probe.py), two processes per setting
cd $GRAFT_REPO_ROOT
run() {
  echo "== $1"
  for k in 1 2; do timeout 400 python tools/pile_alloc_probe.py c3 0 5 3 2>&1 | grep "^context\|rror" | sed 's/(free before: //; s/ GB)//' | awk '{printf "%s ", $0} END {print ""}' | sed 's/context/\n  context/g' | grep context | awk '{print $0}' | tr '\n' ';'; echo; done
}
unset RALA_HIP_PILE_CHUNK_MB RALA_HIP_PILE_CHUNK_ORDER
run "hipMalloc"
export RALA_HIP_PILE_CHUNK_ORDER=0
for mb in 2 64 1024; do export RALA_HIP_PILE_CHUNK_MB=$mb; run "chunks of $mb MB, in order"; done
export RALA_HIP_PILE_CHUNK_ORDER=1
for mb in 2 64 1024; do export RALA_HIP_PILE_CHUNK_MB=$mb; run "chunks of $mb MB, permuted"; done
