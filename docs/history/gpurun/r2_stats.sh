# per-kernel times of a C3 bench run (rocprofv3 --kernel-trace --stats), top 30
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/st
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/st_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/st.log
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/st/*/*kernel_trace.csv
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/st/*/*kernel_stats.csv"):
    for row in list(csv.DictReader(open(f)))[:30]:
        print("%-70s %5s %9.1f us  tot/step %8.1f us" % (row["Name"].replace("rala_hip::(anonymous namespace)::","").replace("rala_hip::","")[:70], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 6e3))
PY
