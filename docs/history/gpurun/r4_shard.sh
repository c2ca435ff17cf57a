# the sharded path's own cost at world 1 (ranks as threads, through RCCL), C3 and C5; 8 ranks sharing the GPU at C3
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04; mkdir -p $OUT
RALA_FORCE_SHARDED=1 python bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r04_c3_bench_sharded_world1.json
RALA_FORCE_SHARDED=1 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>$OUT/c5w1.log | grep "^{" > $OUT/r04_c5_bench_sharded_world1.json
python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r04_c5_bench_1gpu.json
python bench.py --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r04_c3_bench_8ranks_one_gpu.json
python3 - <<'PY'
import json
for f in ("r04_c3_bench_sharded_world1", "r04_c5_bench_sharded_world1", "r04_c5_bench_1gpu", "r04_c3_bench_8ranks_one_gpu"):
    try:
        d = json.load(open("gpurun_out/r04/%s.json" % f))
        print(f, round(d["ms_per_step"], 2), d["transport"], d["rccl_ranks"], {k: round(v, 2) for k, v in d["stage_ms"].items() if k.endswith("_ms")})
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $OUT/c5w1.log
