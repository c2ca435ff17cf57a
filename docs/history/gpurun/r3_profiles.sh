# round 3: everything profiles/ cites, in one call
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 3 > $OUT/r03_c3_bench.json 2> $OUT/bench.log
RALA_FORCE_SHARDED=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r03_c3_bench_sharded_world1.json
python bench.py --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r03_c3_bench_8ranks_one_gpu.json
python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r03_c5_bench_1gpu.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/r03_c3_bench_under_rocprof.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/r03_c3_kernel_stats.csv
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/stats/*/*kernel_trace.csv | head -1) ALL > $OUT/r03_c3_step_trace.txt
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $ROOT/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/stats5.log
cp $(ls $OUT/stats5/*/*kernel_stats.csv | head -1) $OUT/r03_c5_kernel_stats.csv
rm -rf $OUT/stats5
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/write.log
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json
tot = {}
for name in ("fetch", "write"):
    acc = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/r03/%s/*/*counter_collection.csv" % name):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "pile_runs_kernel" in k or "pile_build_annotate" in k:
                short = ("pile_runs_kernel<512>" if "<512" in k else "pile_runs_kernel<1024>" if "<1024" in k else
                         "pile_runs_kernel<2048>" if "<2048" in k else "pile_build_annotate")
                acc[short] += float(row["Counter_Value"])
    tot[name] = dict(acc)
fetch_kb = sum(tot["fetch"].values()); write_kb = sum(tot["write"].values())
out = {"workload": "c3", "kernel": "pile_runs_kernel chain", "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
       "per_kernel_fetch_kb": tot["fetch"], "per_kernel_write_kb": tot["write"],
       "hbm_bytes_per_step": (2.0 * fetch_kb + write_kb) * 1024.0,
       "note": "round 3; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one bench step; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md)"}
json.dump(out, open("gpurun_out/r03/pmc_latest.json", "w"), indent=1)
print(out["hbm_bytes_per_step"])
PY
rm -rf $OUT/fetch $OUT/write
python3 -c "
import json
d=json.load(open('$OUT/r03_c3_bench.json')); print('c3', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['stage_frac']); print(d['stage_ms'])
d=json.load(open('$OUT/r03_c5_bench_1gpu.json')); print('c5', d['ms_per_step'], d['value'], d['roofline']['frac']); print(d['stage_ms'])
d=json.load(open('$OUT/r03_c3_bench_sharded_world1.json')); print('world1', d['ms_per_step']); print(d['stage_ms'])
"
