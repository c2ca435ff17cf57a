# per-kernel times of the pile chain on the long-read probe (factor 2)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/long_prof -- python3 $ROOT/tools/long_read_probe.py 2 > $ROOT/gpurun_out/long_prof.log 2>&1
grep "^x" $ROOT/gpurun_out/long_prof.log
rm -f $ROOT/gpurun_out/long_prof/*/*kernel_trace.csv
python3 - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/long_prof/*/*kernel_stats.csv"):
    for row in list(csv.DictReader(open(f)))[:14]:
        print("%-90s calls %5s avg %10.1f us" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
