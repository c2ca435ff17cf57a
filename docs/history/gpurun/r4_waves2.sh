# pile kernel on the round's final layout: several reads per workgroup (RALA_PILE_WAVES), occupancy (RALA_PILE_EXTRA_LDS), one box
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f tail %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['stage_ms']['tail_host_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2; do
  echo "one read per workgroup : $(run)"
  echo "two                    : $(RALA_PILE_WAVES=2 run)"
  echo "four                   : $(RALA_PILE_WAVES=4 run)"
  echo "6 per SIMD             : $(RALA_PILE_EXTRA_LDS=1088 run)"
done
