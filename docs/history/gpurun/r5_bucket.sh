# round 5: duplicate removal's first pass inside the counting pass (A/B against RALA_DEDUPE_APART=1), the side stream at high priority (c5s)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05f
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_wrap.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python tests/fuzz_parity.py 60 2>&1 | tail -3
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
for k in 1 2; do
q --steps 10 --warmup 2 > $OUT/c3_$k.json
RALA_DEDUPE_APART=1 q --steps 10 --warmup 2 > $OUT/c3_apart_$k.json
done
q --workload c5 --steps 4 --warmup 1 > $OUT/c5.json
RALA_DEDUPE_APART=1 q --workload c5 --steps 4 --warmup 1 > $OUT/c5_apart.json
q --workload c5s --steps 4 --warmup 1 > $OUT/c5s.json
q --workload c3s --steps 10 --warmup 2 > $OUT/c3s.json
for f in c3_1 c3_apart_1 c3_2 c3_apart_2 c5 c5_apart c3s c5s; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), d.get('sensitive_pass',{}).get('ms'), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
