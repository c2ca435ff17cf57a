ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
STOPS=141,142,143,122,123,124,131,125,144,145,126,127,128,77
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $ROOT/gpurun_out/pmc/p3 -- python3 $ROOT/tools/pile_once.py c2 $STOPS > $ROOT/gpurun_out/pmc/p3.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, os
stops = [141,142,143,122,123,124,131,125,144,145,126,127,128,77]
for f in glob.glob("gpurun_out/pmc/p3/*/*counter_collection.csv"):
    per = collections.defaultdict(dict)
    for row in csv.DictReader(open(f)):
        if "pile_runs_kernel<512" in row["Kernel_Name"]:
            d = int(row["Dispatch_Id"]); per[d][row["Counter_Name"]] = per[d].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"]
    print("stop " + " ".join("%14s" % n[3:] for n in names))
    prev = None
    for s, d in zip(stops, sorted(per)):
        cur = [per[d].get(n, 0.0) / 1e5 for n in names]
        print("%4d " % s + " ".join("%14.0f" % x for x in cur))
        if prev: print("  +  " + " ".join("%14.0f" % (x - y) for x, y in zip(cur, prev)))
        prev = cur
for f in glob.glob("gpurun_out/pmc/p3/*/*"):
    if not f.endswith("counter_collection.csv"): os.remove(f)
PY
tail -n 3 gpurun_out/pmc/p3.log
