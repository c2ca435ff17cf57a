# RECORD of a round-4 measurement: the variant it switches on was removed from the tree after the measurement (results in
# DESIGN.md section 4, "Round 4"); the script is kept for what it measured and how.
# duplicate removal beside the bucketing (RALA_DEDUPE_EARLY) instead of beside the pile kernels
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['config']['transitive_pairs']))"; }
for k in 1 2 3; do
  echo "beside the pile kernels: $(run)"
  echo "beside the bucketing   : $(RALA_DEDUPE_EARLY=1 run  # (now the default; RALA_DEDUPE_LATE=1 selects the other))"
done
