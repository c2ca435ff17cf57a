# two reads per workgroup (default now) against one (RALA_PILE_WAVES=1): C3, C5, c3s, one box
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps $2 --warmup 2 $1 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2; do
  echo "c3 two : $(run '' 10)"
  echo "c3 one : $(RALA_PILE_WAVES=1 run '' 10)"
  echo "c5 two : $(run '--workload c5' 4)"
  echo "c5 one : $(RALA_PILE_WAVES=1 run '--workload c5' 4)"
done
echo "c2 two : $(run '--workload c2' 10)"
echo "c2 one : $(RALA_PILE_WAVES=1 run '--workload c2' 10)"
