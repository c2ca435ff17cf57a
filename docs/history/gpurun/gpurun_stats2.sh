ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/st2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/st2 -- python3 $ROOT/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/st2.log
cd $ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/st2/*/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out = []
    for r in rows:
        n = r["Kernel_Name"]
        if "cc_hook" in n or "cc_compress" in n or "death_round" in n:
            out.append("%s %.1f" % ("hook" if "cc_hook" in n else "comp" if "compress" in n else "tdr" if "tail_death" in n else "dr", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print(" | ".join(out[len(out)//2:]))
PY
rm -rf gpurun_out/st2
