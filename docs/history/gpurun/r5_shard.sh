# round 5: the sharded path (bounds scattered once on the sender): parity suites, then world 1 through RCCL and 8 ranks on the one GPU
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fixed_point or golden" 2>&1 | tail -5
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
q --steps 10 --warmup 2 > $OUT/c3_quick.json
RALA_FORCE_SHARDED=1 q --steps 6 --warmup 2 > $OUT/c3_sharded_world1.json || tail -5 $OUT/err.log
q --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 > $OUT/c3_8ranks.json || tail -5 $OUT/err.log
for f in c3_quick c3_sharded_world1 c3_8ranks; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
