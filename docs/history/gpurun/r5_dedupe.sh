# round 5: duplicate removal through the list of marks (parity, shuffled runs in the fuzzer, A/B on the bucketing's time)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c2 or c3" 2>&1 | tail -3
timeout 900 python tests/fuzz_parity.py 100 2>&1 | tail -2
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
for k in 1 2; do
q --steps 10 --warmup 2 > $OUT/c3_$k.json
RALA_DEDUPE_APART=1 q --steps 10 --warmup 2 > $OUT/c3_apart_$k.json
done
q --workload c5 --steps 4 --warmup 1 > $OUT/c5.json
for f in c3_1 c3_apart_1 c3_2 c3_apart_2 c5; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
RALA_HIP_TRACE=1 python bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 0 2>&1 | grep -i "dedupe\|marks" | head
