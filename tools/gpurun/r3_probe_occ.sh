# pile kernel: occupancy sensitivity on one box (RALA_PILE_EXTRA_LDS: 0 = 28 workgroups per compute unit, 1088 = 24, 2624 = 20)
cd $GRAFT_REPO_ROOT
for p in 0 1088 2624 0 1088 2624; do
  echo "extra lds $p: $(RALA_PILE_EXTRA_LDS=$p python bench.py --no-cpu-baseline --no-e2e --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f' % (d['stage_ms']['pile_ms'], d['ms_per_step']))")"
done
