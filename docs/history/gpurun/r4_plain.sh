# first pile kernel: the read's offsets by four loads that do not wait for each other (kPlain, default) against the general
# front end (RALA_PILE_NOT_PLAIN: four round trips in a row through its branches), one box
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps $2 --warmup 2 $1 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2 3 4 5; do
  echo "c3 plain   : $(run '' 12)"
  echo "c3 general : $(RALA_PILE_NOT_PLAIN=1 run '' 12)"
done
for k in 1 2 3; do
echo "c5 plain   : $(run '--workload c5' 4)"
echo "c5 general : $(RALA_PILE_NOT_PLAIN=1 run '--workload c5' 4)"
done
