ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/st
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/st -- python3 $ROOT/bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/st_bench.json 2> $ROOT/gpurun_out/st.log
cd $ROOT
rm -f gpurun_out/st/*/*kernel_trace.csv
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/st/*/*kernel_stats.csv"):
    for row in list(csv.DictReader(open(f)))[:34]:
        print("%-80s %5s %9.1f us  tot/step %8.1f us" % (row["Name"].replace("rala_hip::(anonymous namespace)::","")[:80], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 5e3))
PY
