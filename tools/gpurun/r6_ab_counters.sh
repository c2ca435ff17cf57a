# round 6: the counters of two (or more) variants of the first pile kernel measured INSIDE one process and one profiler pass each
# (tools/pile_ab.py under rocprofv3 --pmc): what differs between processes cancels
# usage: r6_ab_counters.sh "<variants>" <out.json>     (R6_FLAGS: extra compile flags, e.g. the list of cases)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r06c
mkdir -p $OUT
cd $ROOT
touch rala_amd/csrc/pile_runs_kernel.hip
RALA_HIPCC_FLAGS="-DRALA_PILE_AB $R6_FLAGS" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
cd /tmp && export TMPDIR=/tmp
p=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum"; do
  p=$((p+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$p -- python3 $ROOT/tools/pile_ab.py c3 "$1" 2 3 > $OUT/p$p.log 2>&1 || tail -3 $OUT/p$p.log
done
cd $ROOT
python3 - "$2" <<'PY'
import csv, glob, collections, json, re, sys
PROD = re.compile(r"pile_runs_kernel<512u?, false, 0, true, 16384u?, 2u?, true, (\d+)u?>")
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r06c/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        m = PROD.search(row["Kernel_Name"])
        if m:
            per[int(m.group(1))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob("gpurun_out/r06c/p*/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        m = PROD.search(row["Kernel_Name"])
        if m:
            dur[int(m.group(1))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
out = {}
for v in sorted(per):
    m = {k: sum(x) / len(x) for k, x in per[v].items()}
    out[v] = {"kernel_ms_under_counters": sum(dur[v]) / len(dur[v]), "dispatches": len(dur[v]), "per_read": {k: round(x / 1e6, 1) for k, x in m.items()}}
    print("variant", v, "ms %.3f" % out[v]["kernel_ms_under_counters"], json.dumps(out[v]["per_read"]))
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
rm -rf $OUT/p*/
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
