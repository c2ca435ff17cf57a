# pile kernel: occupancy on one box.  RALA_PILE_EXTRA_LDS on the first kernel: 0 = 28 workgroups per compute unit (7 per
# SIMD), 1088 = 24 (6 per SIMD), 2624 = 20 (5 per SIMD).  Round 3: 4.69 / 4.98 / 5.32 ms; an instantiation for 384 events per
# read (5 056 B, 8 per SIMD, 2 927 reads handed on) 4.65 ms - not kept.
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f overflow %d' % (d['stage_ms']['pile_ms'], d['ms_per_step'], d['stage_ms']['pile_overflow_reads']))"; }
for k in 1 2; do
  echo "7 per SIMD: $(run)"
  echo "6 per SIMD: $(RALA_PILE_EXTRA_LDS=1088 run)"
  echo "5 per SIMD: $(RALA_PILE_EXTRA_LDS=2624 run)"
done
