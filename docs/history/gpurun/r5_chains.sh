# round 5: dependent round trips flattened in death_decide, sens_records, sens_trim, sens_bridge: the suite, fuzzers, timings
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05z
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 900 python tests/fuzz_parity.py 100 70000 2>&1 | tail -1
timeout 900 python tests/fuzz_sharded.py 40 80000 2>&1 | tail -1
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
q --steps 10 --warmup 2 > $OUT/c3.json
q --workload c5 --steps 4 --warmup 1 > $OUT/c5.json
q --workload c3s --steps 6 --warmup 2 > $OUT/c3s.json
q --workload c5s --steps 3 --warmup 1 > $OUT/c5s.json
for f in c3 c5 c3s c5s; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), round(d['roofline']['frac'],3), {k: round(v,3) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
