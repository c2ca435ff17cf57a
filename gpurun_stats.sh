ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/st
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/st -- python3 $ROOT/tools/pile_once.py c3 99,99 > $ROOT/gpurun_out/st.log 2>&1
cd $ROOT
rm -f gpurun_out/st/*/*kernel_trace.csv
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/st/*/*kernel_stats.csv"):
    for row in list(csv.DictReader(open(f)))[:16]:
        print("%-90s %5s %10.1f us" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
