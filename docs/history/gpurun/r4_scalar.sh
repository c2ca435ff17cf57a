# RECORD of a round-4 measurement: the variant it switches on was removed from the tree after the measurement (results in
# DESIGN.md section 4, "Round 4"); the script is kept for what it measured and how.
# first pile kernel, plain front end: the read's offsets by the scalar unit (RALA_PILE_SCALAR_META) against the vector path
cd $GRAFT_REPO_ROOT
RALA_PILE_SCALAR_META=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --no-e2e --steps $2 --warmup 2 $1 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2 3 4; do
  echo "c3 vector : $(run '' 12)"
  echo "c3 scalar : $(RALA_PILE_SCALAR_META=1 run '' 12)"
done
for k in 1 2; do
echo "c5 vector : $(run '--workload c5' 4)"
echo "c5 scalar : $(RALA_PILE_SCALAR_META=1 run '--workload c5' 4)"
echo "c5 general: $(RALA_PILE_NOT_PLAIN=1 run '--workload c5' 4)"
done
