ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/prof
cd $ROOT && python bench.py --steps 5 --warmup 1 > gpurun_out/prof/bench_default.json 2> gpurun_out/prof/bench_default.log; cut -c1-1500 gpurun_out/prof/bench_default.json
RALA_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/prof/bench_sharded_world1.json
python -c "import json; d=json.load(open('gpurun_out/prof/bench_sharded_world1.json')); print('sharded(world=1):', d['value'], d['ms_per_step'], d['stage_ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof/stats -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof/bench_under_rocprof.json 2> $ROOT/gpurun_out/prof/stats.log
rm -f $ROOT/gpurun_out/prof/stats/*/*kernel_trace.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/prof/fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/prof/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/prof/write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/prof/write.log
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json
tot = {}
for name in ("fetch", "write"):
    acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in glob.glob("gpurun_out/prof/%s/*/*counter_collection.csv" % name):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "pile_runs_kernel" in k or "pile_build_annotate" in k:
                short = ("pile_runs_kernel<512>" if "<512" in k else "pile_runs_kernel<1024>" if "<1024" in k else
                         "pile_runs_kernel<2048>" if "<2048" in k else "pile_build_annotate")
                acc[short] += float(row["Counter_Value"]); cnt[short] += 1
    tot[name] = dict(acc)
    print(name, dict(acc), dict(cnt))
fetch_kb = sum(tot["fetch"].values()); write_kb = sum(tot["write"].values())
out = {"workload": "c3", "kernel": "pile_runs_kernel chain", "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
       "per_kernel_fetch_kb": tot["fetch"], "per_kernel_write_kb": tot["write"],
       "hbm_bytes_per_step": (2.0 * fetch_kb + write_kb) * 1024.0,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one bench step; FETCH_SIZE doubled (gfx950); see profiles/r01_c3_pmc_pile_kernel.md"}
json.dump(out, open("gpurun_out/prof/pmc_latest.json", "w"), indent=1)
print(out["hbm_bytes_per_step"])
PY
rm -rf gpurun_out/prof/fetch gpurun_out/prof/write
