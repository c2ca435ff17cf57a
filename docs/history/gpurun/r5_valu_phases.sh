# round 5: vector / scalar / LDS instructions of the pile kernel per phase (the diagnostic instantiation leaving after phase k,
# without the row stores; c2 = 100 k reads, values per read) - where the 1 841 vector instructions of a read go
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STOPS=141,142,143,122,123,124,131,125,144,145,126,127,128,177,99
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d $OUT/p -- python3 $ROOT/tools/pile_once.py c2 $STOPS > $OUT/p.log 2>&1 || tail -3 $OUT/p.log
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
stops = "141,142,143,122,123,124,131,125,144,145,126,127,128,177,99".split(",")
for f in glob.glob("gpurun_out/r05v/p/*/*counter_collection.csv"):
    per = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "pile_runs_kernel" in k and "512" in k and ", true" in k.split("512")[1][:40]:
            d = int(row["Dispatch_Id"])
            per.setdefault(d, collections.defaultdict(float))[row["Counter_Name"]] += float(row["Counter_Value"])
            per[d]["name"] = k[:60]
    prev = collections.defaultdict(float)
    for (d, c), s in zip(per.items(), stops):
        print("stop %4s" % s, " ".join("%s=%.0f (+%.0f)" % (n[3:], c[n] / 1e5, (c[n] - prev[n]) / 1e5) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")), c["name"][-40:])
        prev = c
PY
rm -rf $OUT/p
