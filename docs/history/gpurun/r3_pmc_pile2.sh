# round 3: the diagnostic instantiation with (99 is the product; 98 = diag, everything) and without (77) the row stores: where does a wavefront's time go?
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/pmc4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $ROOT/gpurun_out/pmc4/a -- python3 $ROOT/tools/pile_once.py c3 77,77,98,98,99,99 > $ROOT/gpurun_out/pmc4/a.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc4/a/*/*counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}; name = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "pile_runs_kernel<512" in k and ", true, 16384" in k:
            d = int(row["Dispatch_Id"]); per[d][row["Counter_Name"]] += float(row["Counter_Value"])
            dur[d] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3; name[d] = "diag" if "<512u, true" in k else "product"
    for d in sorted(per):
        print(name[d], "%.0f us" % dur[d], " ".join("%s=%.0f" % (n[3:], v / 1e6) for n, v in sorted(per[d].items())))
PY
grep stop gpurun_out/pmc4/a.log
rm -rf gpurun_out/pmc4/a
