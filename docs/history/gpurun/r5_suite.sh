# round 5: the whole GPU suite, then the quick bench lines (C3, c3s, c5s)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
q --steps 10 --warmup 2 > $OUT/c3.json
RALA_CLASSIFY_NO_VEC=1 q --steps 10 --warmup 2 > $OUT/c3_novec.json
q --workload c3s --steps 10 --warmup 2 > $OUT/c3s.json
q --workload c5s --steps 4 --warmup 1 > $OUT/c5s.json
for f in c3 c3_novec c3s c5s; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), d.get('sensitive_pass',{}).get('ms'), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
