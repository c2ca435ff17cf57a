# round 6: small chunks mapped in a permuted order (thousands of independent placements instead of twenty): the spread, the two
# kinds of row stores on them, and what the mapping costs (set_reads allocates the rows)
cd $GRAFT_REPO_ROOT
run() {
  echo "== $1"
  for k in 1 2; do timeout 400 python tools/pile_alloc_probe.py ${2:-c3} 0,262144 4 3 2>&1 | grep "^context\|rror" | sed 's/(free before: [0-9.]* GB, //; s/)//' | cut -c1-150; done
}
export RALA_HIP_PILE_CHUNK_ORDER=1
for mb in 2 16; do export RALA_HIP_PILE_CHUNK_MB=$mb; run "chunks of $mb MB, permuted"; done
export RALA_HIP_PILE_CHUNK_ORDER=0 RALA_HIP_PILE_CHUNK_MB=1024
run "chunks of 1024 MB, in order"
