# RECORD of a round-4 measurement: the variant it switches on was removed from the tree after the measurement (results in
# DESIGN.md section 4, "Round 4"); the script is kept for what it measured and how.
# pile kernel on one box: the kernel's register count alone (variant 5 = variant 1 + a clobber of v71: 72 registers instead of
# 59, seven wavefronts per SIMD by registers as well as by LDS)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2; do
  for v in 0 8; do echo "variant $v : $(RALA_PILE_EXPAND_OLD=$v run)"; done
done
