# round 5, first call: this round's box baseline (C3 quick, sharded world 1, 8 ranks on the one GPU) and the memory-side
# counters of the product pile kernel beside those of the row-fill microbenchmarks
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05a
mkdir -p $OUT $ROOT/tools/_bin
cd $ROOT
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | grep '^{'; }
q --steps 10 --warmup 2 > $OUT/c3_quick.json
RALA_FORCE_SHARDED=1 q --steps 6 --warmup 2 > $OUT/c3_sharded_world1.json
q --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 > $OUT/c3_8ranks.json
hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fill_bench3 tools/fill_bench3.hip 2>$OUT/fill3_build.log
hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fill_bench4 tools/fill_bench4.hip 2>$OUT/fill4_build.log
tools/_bin/fill_bench3 > $OUT/fill_bench3.txt 2>&1
tools/_bin/fill_bench4 > $OUT/fill_bench4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum" "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL" "TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum TCC_EA0_WR_UNCACHED_32B_sum" "TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum"; do
  tag=$(echo $set | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_pile_$tag -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/pmc_pile_$tag.log
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_fill3_$tag -- $ROOT/tools/_bin/fill_bench3 > /dev/null 2> $OUT/pmc_fill3_$tag.log
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("gpurun_out/r05a/pmc_*")):
    if not os.path.isdir(d): continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        if rows: out.setdefault("_columns", list(rows[0].keys()))
        for row in rows:
            k = row["Kernel_Name"]
            if "pile_runs_kernel<512" in k and "16384" in k or "row_fill" in k or "Fill" in k or "fill" in k:
                per[k[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out[os.path.basename(d)] = {k: {c: {"calls": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()} for k, cs in per.items()}
json.dump(out, open("gpurun_out/r05a/pmc_summary.json", "w"), indent=1)
PY
# keep the raw csv of the per-instance pass small: drop everything else
for d in $OUT/pmc_*; do [ -d $d ] && find $d -name '*.csv' -size +20M -delete; done
du -sh $OUT
for f in c3_quick c3_sharded_world1 c3_8ranks; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
cat $OUT/fill_bench3.txt
