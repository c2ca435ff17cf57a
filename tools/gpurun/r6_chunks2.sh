# round 6: the rows' buffer as chunks (hipMemCreate / hipMemMap): more processes, larger chunks, and what the allocation costs
cd $GRAFT_REPO_ROOT
run() {
  echo "== $1"
  for k in 1 2 3; do timeout 400 python tools/pile_alloc_probe.py c3 0 5 3 2>&1 | grep "^context\|rror" | sed 's/context \([0-9]\) (free before: [0-9.]* GB): var 0: [0-9.]* \([0-9.]*\) \([0-9.]*\)/\2/' | tr '\n' ' '; echo; done
}
unset RALA_HIP_PILE_CHUNK_MB RALA_HIP_PILE_CHUNK_ORDER
run "hipMalloc"
export RALA_HIP_PILE_CHUNK_ORDER=0
for mb in 1024 4096; do export RALA_HIP_PILE_CHUNK_MB=$mb; run "chunks of $mb MB, in order"; done
export RALA_HIP_PILE_CHUNK_ORDER=1
for mb in 256 1024; do export RALA_HIP_PILE_CHUNK_MB=$mb; run "chunks of $mb MB, permuted"; done
unset RALA_HIP_PILE_CHUNK_MB RALA_HIP_PILE_CHUNK_ORDER
run "hipMalloc again"
