# round 6: everything profiles/r06_* cites, in one call on the final build
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 3 > $OUT/r06_c3_bench.json 2> $OUT/bench.log
python bench.py --workload c3s --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r06_c3s_bench.json
python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r06_c5_bench_1gpu.json
python bench.py --workload c5s --steps 4 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r06_c5s_bench_1gpu.json
RALA_FORCE_SHARDED=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r06_c3_bench_sharded_world1.json
RALA_FORCE_SHARDED=1 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r06_c5_bench_sharded_world1.json
python bench.py --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $OUT/r06_c3_bench_8ranks_one_gpu.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/r06_c3_bench_under_rocprof.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/r06_c3_kernel_stats.csv
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/stats/*/*kernel_trace.csv | head -1) ALL > $OUT/r06_c3_step_trace.txt
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statss -- python3 $ROOT/bench.py --workload c3s --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/statss.log
cp $(ls $OUT/statss/*/*kernel_stats.csv | head -1) $OUT/r06_c3s_kernel_stats.csv
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/statss/*/*kernel_trace.csv | head -1) ALL > $OUT/r06_c3s_step_trace.txt
rm -rf $OUT/statss
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $ROOT/bench.py --workload c5s --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/stats5.log
cp $(ls $OUT/stats5/*/*kernel_stats.csv | head -1) $OUT/r06_c5s_kernel_stats.csv
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/stats5/*/*kernel_trace.csv | head -1) ALL > $OUT/r06_c5s_step_trace.txt
rm -rf $OUT/stats5
# HBM traffic of the pile chain: separate PMC passes (MI355X_MICROARCH.md: FETCH_SIZE counts half of wide streaming reads on gfx950)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/write.log
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json
def short(k):
    k = k.replace("rala_hip::", "").replace("(anonymous namespace)::", "")
    if k.startswith("void "): k = k[5:]
    if "pile_runs_kernel" in k:
        return "pile_runs_kernel<%s>" % k.split("<")[1].split(",")[0].rstrip("u")
    return k.split("(")[0].split("<")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for name in ("fetch", "write"):
    for f in glob.glob("gpurun_out/r06/%s/*/*counter_collection.csv" % name):
        for row in csv.DictReader(open(f)):
            per[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
pile = [k for k in per if k.startswith("pile_runs_kernel") or k == "pile_build_annotate"]
fetch_kb = sum(per[k]["FETCH_SIZE"] for k in pile); write_kb = sum(per[k]["WRITE_SIZE"] for k in pile)
out = {"workload": "c3", "kernel": "pile_runs_kernel chain", "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
       "per_kernel_fetch_kb": {k: per[k]["FETCH_SIZE"] for k in pile}, "per_kernel_write_kb": {k: per[k]["WRITE_SIZE"] for k in pile},
       "hbm_bytes_per_step": (2.0 * fetch_kb + write_kb) * 1024.0,
       "note": "round 6; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one bench step; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md)"}
json.dump(out, open("gpurun_out/r06/pmc_latest.json", "w"), indent=1)
print("pile chain HBM bytes per step", out["hbm_bytes_per_step"])
PY
rm -rf $OUT/fetch $OUT/write
# the pile kernel's issue model (product instantiation beside round 5's kernel, one process) and its sensitivity to added work
R6_FLAGS="'-DRALA_PILE_AB_CASES=X(8195)'" bash tools/gpurun/r6_ab_counters.sh 0,8195 gpurun_out/r06/r06_c3_pile_issue_model.json > $OUT/issue_model.log 2>&1
# (0 = the product: non-temporal row stores, the rows in mapped chunks; 262144 = plain stores; 65536 + k = the product with work added)
R6_PROCS=2 R6_ROUNDS=5 R6_STEPS=4 R6_FLAGS="'-DRALA_PILE_AB_CASES=X(65792) X(66816) X(66048) X(66304) X(66560) X(73728) X(8195) X(65537) X(65538)'" bash tools/gpurun/r6_ab_inproc.sh 0,262144,65792,66816,66048,66304,66560,73728,8195,65537,65538 > $OUT/r06_c3_pile_sensitivity.txt 2>&1
python3 -c "
import json
for f in ('r06_c3_bench','r06_c3s_bench','r06_c5_bench_1gpu','r06_c5s_bench_1gpu','r06_c3_bench_sharded_world1','r06_c5_bench_sharded_world1','r06_c3_bench_8ranks_one_gpu'):
    try:
        d=json.load(open('$OUT/'+f+'.json')); print(f, round(d['ms_per_step'],2), round(d['value']/1e9,2), round(d['roofline']['frac'],3), d.get('sensitive_pass',{}).get('ms'), (d.get('result_check') or {}).get('ok'))
    except Exception as e: print(f, 'failed', e)
"
