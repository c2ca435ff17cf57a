# round 6, final build: the bench line (without its host-side figures) from ten processes after each other on one box - where the
# pile chain lands from process to process now that the rows lie in mapped chunks and are stored non-temporally
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('process $i: %.2f ms = %.2f G overlaps/s, pile chain %.3f ms, frac %.3f, bucketing %.3f, check %s' % (d['ms_per_step'], d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'], d['stage_ms']['bucket_ms'], d['result_check']['ok']))"
done
