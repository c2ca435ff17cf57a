# round 5: classify with the killers noted in LDS and the comparisons in integers: parity, fuzzer, C3 / C5 timings
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 900 python tests/fuzz_parity.py 80 2>&1 | tail -1
timeout 600 python tests/fuzz_sharded.py 20 2>&1 | tail -1
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
q --steps 10 --warmup 2 > $OUT/c3_1.json
q --steps 10 --warmup 2 > $OUT/c3_2.json
q --workload c5 --steps 4 --warmup 1 > $OUT/c5.json
for f in c3_1 c3_2 c5; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), round(d['roofline']['frac'],3), {k: round(v,3) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
