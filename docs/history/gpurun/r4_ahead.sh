# EXPERIMENT, not in the tree (round 4): expansion with the LDS reads of two groups under way before the first value is looked
# at against one group at a time (RALA_PILE_AHEAD1), one box: 4.11 - 4.14 ms either way
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --no-e2e --steps $2 --warmup 2 $1 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2 3 4; do
  echo "c3 two ahead : $(run '' 12)"
  echo "c3 one       : $(RALA_PILE_AHEAD1=1 run '' 12)"
done
for k in 1 2; do
echo "c5 two ahead : $(run '--workload c5' 4)"
echo "c5 one       : $(RALA_PILE_AHEAD1=1 run '--workload c5' 4)"
done
