# round 5: every rank tokenises its own byte range (tests), the CLI over several ranks with it, C5 sharded at world 1, the end-to-end figure over 8 ranks
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "several_ranks or device_tokeniser" 2>&1 | tail -15
q() { python bench.py --no-cpu-baseline "$@" 2>$OUT/err.log | grep '^{'; }
q --no-e2e --workload c5 --steps 3 --warmup 1 > $OUT/c5_quick.json
RALA_FORCE_SHARDED=1 q --no-e2e --workload c5 --steps 3 --warmup 1 > $OUT/c5_sharded_world1.json || tail -5 $OUT/err.log
q --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 > $OUT/c3_8ranks_e2e.json || tail -5 $OUT/err.log
for f in c5_quick c5_sharded_world1 c3_8ranks_e2e; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v}); print(d.get('end_to_end_from_paf'))"; done
