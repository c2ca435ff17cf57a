# round 5: the step through RCCL with a world of one against the unsharded one, alternating in one box (C3 three times, C5 once)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05w
mkdir -p $OUT
cd $ROOT
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
for k in 1 2 3; do
q --steps 8 --warmup 2 > $OUT/c3_$k.json
RALA_FORCE_SHARDED=1 q --steps 8 --warmup 2 > $OUT/c3_sharded_$k.json
done
q --workload c5 --steps 4 --warmup 1 > $OUT/c5.json
RALA_FORCE_SHARDED=1 q --workload c5 --steps 4 --warmup 1 > $OUT/c5_sharded.json
for f in c3_1 c3_sharded_1 c3_2 c3_sharded_2 c3_3 c3_sharded_3 c5 c5_sharded; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v and k in ('bucket_ms','pile_ms','emit_ms','owner_ms','construct_ms','owner_pile_ms','owner_bucket_ms','classify_ms','total_ms')})"; done
