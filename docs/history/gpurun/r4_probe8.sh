# RECORD of a round-4 measurement: the variant it switches on was removed from the tree after the measurement (results in
# DESIGN.md section 4, "Round 4"); the script is kept for what it measured and how.
# TIMING ONLY (the reads with 385 .. 512 events are processed by nobody): what an eighth wavefront per SIMD would bring the
# first pile kernel if its LDS footprint fell to 5 120 B with the event cap unchanged.  RALA_PILE_PROBE_CAP: the 512-event
# kernel skips those reads as well (the same work at seven wavefronts); + RALA_PILE_PROBE8: the 384-event instantiation, eight.
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f' % (d['stage_ms']['pile_ms'], d['ms_per_step']))"; }
for k in 1 2 3; do
  echo "7 per SIMD, all reads      : $(run)"
  echo "7 per SIMD, 99.7 % of them : $(RALA_PILE_PROBE_CAP=1 run)"
  echo "8 per SIMD, 99.7 % of them : $(RALA_PILE_PROBE_CAP=1 RALA_PILE_PROBE8=1 run)"
done
