# A/B inside one call: bench lines with and without an option / environment variable
cd $GRAFT_REPO_ROOT
line() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1: ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items() if k.endswith('_ms')})"; }
line base
RALA_BENCH_OPTIONS="$1" line "$1"
line base
RALA_BENCH_OPTIONS="$1" line "$1"
