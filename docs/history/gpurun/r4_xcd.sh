# RECORD of a round-4 measurement: the variant it switches on was removed from the tree after the measurement (results in
# DESIGN.md section 4, "Round 4"); the script is kept for what it measured and how.
# pile kernel, one workgroup per read: every XCD writes its own contiguous eighth of the rows (RALA_PILE_XCD_ROWS), one box
cd $GRAFT_REPO_ROOT
RALA_PILE_XCD_ROWS=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 $1 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2 3; do
  echo "rows as they come : $(run)"
  echo "XCD-contiguous    : $(RALA_PILE_XCD_ROWS=1 run)"
done
echo "c5 rows as they come : $(run '--workload c5')"
echo "c5 XCD-contiguous    : $(RALA_PILE_XCD_ROWS=1 run '--workload c5')"
