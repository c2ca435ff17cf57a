# round 5: everything profiles/ cites, in one call
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 3 > $OUT/r05_c3_bench.json 2> $OUT/bench.log
python bench.py --workload c3s --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r05_c3s_bench.json
python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r05_c5_bench_1gpu.json
python bench.py --workload c5s --steps 4 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r05_c5s_bench_1gpu.json
RALA_FORCE_SHARDED=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r05_c3_bench_sharded_world1.json
RALA_FORCE_SHARDED=1 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r05_c5_bench_sharded_world1.json
python bench.py --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $OUT/r05_c3_bench_8ranks_one_gpu.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/r05_c3_bench_under_rocprof.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/r05_c3_kernel_stats.csv
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/stats/*/*kernel_trace.csv | head -1) ALL > $OUT/r05_c3_step_trace.txt
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statss -- python3 $ROOT/bench.py --workload c3s --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/statss.log
cp $(ls $OUT/statss/*/*kernel_stats.csv | head -1) $OUT/r05_c3s_kernel_stats.csv
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/statss/*/*kernel_trace.csv | head -1) ALL > $OUT/r05_c3s_step_trace.txt
rm -rf $OUT/statss
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $ROOT/bench.py --workload c5s --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/stats5.log
cp $(ls $OUT/stats5/*/*kernel_stats.csv | head -1) $OUT/r05_c5s_kernel_stats.csv
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/stats5/*/*kernel_trace.csv | head -1) ALL > $OUT/r05_c5s_step_trace.txt
rm -rf $OUT/stats5
# HBM traffic: separate PMC passes (MI355X_MICROARCH.md: FETCH_SIZE counts half of wide streaming reads on gfx950)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/write.log
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/wrreq -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/wrreq.log
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json
N_OVL = 50858245
def short(k):
    k = k.replace("rala_hip::", "").replace("(anonymous namespace)::", "")
    if k.startswith("void "): k = k[5:]
    if "pile_runs_kernel" in k:
        return "pile_runs_kernel<%s>" % k.split("<")[1].split(",")[0].rstrip("u")
    return k.split("(")[0].split("<")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for name in ("fetch", "write", "wrreq"):
    for f in glob.glob("gpurun_out/r05/%s/*/*counter_collection.csv" % name):
        for row in csv.DictReader(open(f)):
            per[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
pile = [k for k in per if k.startswith("pile_runs_kernel") or k == "pile_build_annotate"]
fetch_kb = sum(per[k]["FETCH_SIZE"] for k in pile); write_kb = sum(per[k]["WRITE_SIZE"] for k in pile)
out = {"workload": "c3", "kernel": "pile_runs_kernel chain", "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
       "per_kernel_fetch_kb": {k: per[k]["FETCH_SIZE"] for k in pile}, "per_kernel_write_kb": {k: per[k]["WRITE_SIZE"] for k in pile},
       "hbm_bytes_per_step": (2.0 * fetch_kb + write_kb) * 1024.0,
       "note": "round 5; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one bench step; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md)"}
json.dump(out, open("gpurun_out/r05/pmc_latest.json", "w"), indent=1)
# the bucketing's kernels: bytes moved against the 40 B per overlap = 2.03 GB the stage moves algorithmically
bk = ["group_count_kernel", "group_count_dedupe_kernel", "dedupe_fix_kernel", "dedupe_fix_list_kernel", "layout_kernel", "l1_scatter_kernel", "l2_scatter_kernel", "group_query_sum_kernel", "group_event_base_kernel",
      "final_kernel", "query_side_kernel"]
rows = []
tot_f = tot_w = 0.0
for k in bk:
    if k not in per: continue
    f = 2.0 * per[k]["FETCH_SIZE"] * 1024.0; w = per[k]["WRITE_SIZE"] * 1024.0
    tot_f += f; tot_w += w
    rows.append({"kernel": k, "fetched_bytes_x2_corrected": f, "written_bytes": w, "wrreq_per_overlap": per[k]["TCC_EA0_WRREQ_sum"] / N_OVL,
                 "wrreq_64B_per_overlap": per[k]["TCC_EA0_WRREQ_64B_sum"] / N_OVL, "atomics_per_overlap": per[k]["TCC_EA0_ATOMIC_sum"] / N_OVL})
json.dump({"workload": "c3", "n_overlaps": N_OVL, "algorithmic_bytes": 40.0 * N_OVL, "fetched_bytes": tot_f, "written_bytes": tot_w,
           "moved_over_algorithmic": (tot_f + tot_w) / (40.0 * N_OVL), "kernels": rows,
           "note": "round 5; partitioned bucketing (the default path); one rocprofv3 --pmc pass per counter group over one bench step"},
          open("gpurun_out/r05/r05_c3_pmc_bucket_partitioned.json", "w"), indent=1)
print("pile chain HBM bytes per step", out["hbm_bytes_per_step"], "bucketing moved/algorithmic", (tot_f + tot_w) / (40.0 * N_OVL))
PY
rm -rf $OUT/fetch $OUT/write $OUT/wrreq
python3 -c "
import json
for f in ('r05_c3_bench','r05_c3s_bench','r05_c5_bench_1gpu','r05_c5s_bench_1gpu','r05_c3_bench_sharded_world1','r05_c5_bench_sharded_world1','r05_c3_bench_8ranks_one_gpu'):
    try:
        d=json.load(open('$OUT/'+f+'.json')); print(f, round(d['ms_per_step'],2), round(d['value']/1e9,2), round(d['roofline']['frac'],3), d.get('sensitive_pass',{}).get('ms'))
    except Exception as e: print(f, 'failed', e)
"
