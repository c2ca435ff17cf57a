# persistent pile kernel: grid sweep (static shares: more, shorter-lived workgroups = closer to dynamic balancing)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f overflow %d tr %d' % (d['stage_ms']['pile_ms'], d['ms_per_step'], d['stage_ms']['pile_overflow_reads'], d['config']['transitive_pairs']))"; }
echo "one per read   : $(run)"
for g in 14336 28672 57344 114688 229376 458752; do
  echo "persist $g : $(RALA_PILE_PERSIST2=$g run)"
done
echo "one per read   : $(run)"
