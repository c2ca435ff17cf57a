# round 5: the query side from the columns against through the slots, C3 twice and C5, one box
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05k
mkdir -p $OUT
cd $ROOT
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
for k in 1 2; do
q --steps 10 --warmup 2 > $OUT/direct_$k.json
RALA_QUERY_THROUGH_SLOTS=1 q --steps 10 --warmup 2 > $OUT/slots_$k.json
done
q --workload c5 --steps 4 --warmup 1 > $OUT/c5_direct.json
RALA_QUERY_THROUGH_SLOTS=1 q --workload c5 --steps 4 --warmup 1 > $OUT/c5_slots.json
for f in direct_1 slots_1 direct_2 slots_2 c5_direct c5_slots; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), round(d['roofline']['frac'],3), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
