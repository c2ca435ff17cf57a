# round 6: the bucketing's kernels at C5 with 4 096 / 8 192 / 16 384 reads per first-level partition (one process each, kernel trace)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for m in 12 13 14; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r06ps/m$m -o run -- python3 tools/bucket_shift_ab.py c5 $m 2 > gpurun_out/r06ps/m$m.log 2>&1
  f=$(ls gpurun_out/r06ps/m$m/*/run_kernel_stats.csv gpurun_out/r06ps/m$m/run_kernel_stats.csv 2>/dev/null | head -1)
  echo "== shift $m"; tail -1 gpurun_out/r06ps/m$m.log
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("group_count", "layout", "l1_scatter", "l2_scatter", "final_kernel", "query_side", "group_query", "group_event")):
        print("  %-60s calls %4s avg %9.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
