cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_host_api.py -x -q 2>&1 | tail -3
RALA_HIP_TRACE=1 python bench.py --workload c3 --steps 1 --warmup 2 --no-cpu-baseline 2>&1 | grep "trace" | tail -40
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/st_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/st.log
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/st/*/*kernel_trace.csv
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/st/*/*kernel_stats.csv"):
    for row in list(csv.DictReader(open(f)))[:60]:
        print("%-70s %5s %9.1f us  tot/step %8.1f us" % (row["Name"].replace("rala_hip::(anonymous namespace)::","").replace("rala_hip::","")[:70], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 6e3))
PY
