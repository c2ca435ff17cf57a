# pile rows on 16-byte (product), 128-byte, 1 KB, 4 KB boundaries (RALA_PILE_ROW_ALIGN in elements), one box
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2; do
  for v in 8 64 512 2048; do echo "align $v : $(RALA_PILE_ROW_ALIGN=$v run)"; done
done
