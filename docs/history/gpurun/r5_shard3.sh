# round 5: the sender's count with duplicate removal's first pass inside (as on one GPU): sharded tests, the sharded fuzzer,
# the step through RCCL with a world of one against the unsharded one (C3 twice, C5), 8 ranks on the one GPU
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05y
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_cli.py tests/test_gpu_ingest.py tests/test_gpu_bench.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 900 python tests/fuzz_sharded.py 60 2>&1 | tail -1
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
for k in 1 2; do
q --steps 8 --warmup 2 > $OUT/c3_$k.json
RALA_FORCE_SHARDED=1 q --steps 8 --warmup 2 > $OUT/c3_sharded_$k.json
done
q --workload c5 --steps 4 --warmup 1 > $OUT/c5.json
RALA_FORCE_SHARDED=1 q --workload c5 --steps 4 --warmup 1 > $OUT/c5_sharded.json
q --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 > $OUT/c3_8ranks.json
for f in c3_1 c3_sharded_1 c3_2 c3_sharded_2 c5 c5_sharded c3_8ranks; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v and k in ('bucket_ms','pile_ms','emit_ms','owner_ms','construct_ms','owner_pile_ms','owner_bucket_ms','classify_ms','total_ms')})"; done
