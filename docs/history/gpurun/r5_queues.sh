# round 5: the sharded step (world 1, RCCL) with the copy stream made only when asked for, and with more hardware queues than
# the four a process has by default (GPU_MAX_HW_QUEUES): do a context's streams share queues?
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05x
mkdir -p $OUT
cd $ROOT
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
for k in 1 2; do
q --steps 8 --warmup 2 > $OUT/c3_$k.json
RALA_FORCE_SHARDED=1 q --steps 8 --warmup 2 > $OUT/c3_sharded_$k.json
GPU_MAX_HW_QUEUES=8 RALA_FORCE_SHARDED=1 q --steps 8 --warmup 2 > $OUT/c3_sharded_q8_$k.json
GPU_MAX_HW_QUEUES=8 q --steps 8 --warmup 2 > $OUT/c3_q8_$k.json
done
GPU_MAX_HW_QUEUES=8 q --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 > $OUT/c3_8ranks_q8.json
q --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 > $OUT/c3_8ranks.json
for f in c3_1 c3_sharded_1 c3_sharded_q8_1 c3_q8_1 c3_2 c3_sharded_2 c3_sharded_q8_2 c3_q8_2 c3_8ranks c3_8ranks_q8; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v and k in ('bucket_ms','pile_ms','emit_ms','owner_ms','construct_ms','owner_pile_ms','owner_bucket_ms','classify_ms','total_ms')})"; done
