# round 5: the first pile kernel reads a read's query side straight from the file's columns (parity, fuzzers, A/B against
# RALA_QUERY_THROUGH_SLOTS=1 = query_side_kernel for every overlap as before)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05i
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python tests/fuzz_parity.py 100 2>&1 | tail -2
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
for k in 1 2; do
q --steps 10 --warmup 2 > $OUT/c3_$k.json
RALA_QUERY_THROUGH_SLOTS=1 q --steps 10 --warmup 2 > $OUT/c3_slots_$k.json
done
q --workload c5 --steps 4 --warmup 1 > $OUT/c5.json
RALA_QUERY_THROUGH_SLOTS=1 q --workload c5 --steps 4 --warmup 1 > $OUT/c5_slots.json
q --workload c3s --steps 6 --warmup 2 > $OUT/c3s.json
q --workload c5s --steps 3 --warmup 1 > $OUT/c5s.json
for f in c3_1 c3_slots_1 c3_2 c3_slots_2 c5 c5_slots c3s c5s; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), d['roofline']['frac'], {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
