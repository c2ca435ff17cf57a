# round 6: the bench line itself with the rows' buffer from hipMalloc and as chunks of 1 GB mapped in order, processes alternating
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s: %.2f ms, pile chain %.3f ms, frac %.3f, bucketing %.3f, check %s' % ('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['stage_ms']['bucket_ms'], d['result_check']['ok']))"; }
for i in 1 2 3; do
  export RALA_HIP_PILE_CHUNK_MB=0
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | line "hipMalloc $i"
  export RALA_HIP_PILE_CHUNK_MB=1024
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | line "chunks    $i"
done
export RALA_HIP_PILE_CHUNK_MB=0

export RALA_HIP_PILE_CHUNK_MB=1024

