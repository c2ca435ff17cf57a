ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
RALA_HIP_TRACE=1 python bench.py --workload c5s --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/c5s_trace.log
grep "rep:\|sens\|chain" $OUT/c5s_trace.log | tail -22
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr5 -- python3 $ROOT/bench.py --workload c5s --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/tr5.log
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ.get("OUT", "") + "/tr5/*/*kernel_trace.csv") or glob.glob("/root/repo/gpurun_out/r04/tr5/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    k = r["Kernel_Name"]
    if "pile_runs_kernel" in k and (", 1, " in k or ", 2, " in k) or "pile_repeats" in k or "list_targets" in k or "list_members" in k or "sens_" in k:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is None: t0 = s
        g = r.get("Grid_Size", r.get("Grid_Size_X", "?")); wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
        print("%9.3f ms +%8.3f ms grid %s wg %s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, g, wg, k.replace("rala_hip::", "")[:70]))
PY
rm -rf $OUT/tr5
