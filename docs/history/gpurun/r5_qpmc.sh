# round 5: SQ counters of the plain pile kernel, query side from the columns against through the slots (c2: 100 k reads,
# values per read)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/pile_once.py c2 99,99 > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log
}
for mode in direct slots; do
  if [ $mode = slots ]; then export RALA_QUERY_THROUGH_SLOTS=1; else unset RALA_QUERY_THROUGH_SLOTS; fi
  pass ${mode}_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY
  pass ${mode}_b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH
  pass ${mode}_c SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS SQ_LDS_BANK_CONFLICT
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for mode in ("direct", "slots"):
  for name in "abc":
    for f in glob.glob("gpurun_out/r05q/%s_%s/*/*counter_collection.csv" % (mode, name)):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        dur = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "pile_runs_kernel" in k and "512" in k and "true" in k:
                per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
                dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3 if "End_Timestamp" in row else 0
        for d in sorted(per)[-1:]:
            print(mode, name, "(%.0f us)" % dur.get(d, 0), " ".join("%s=%.0f" % (n[3:] if n.startswith("SQ_") else n, v / 1e5) for n, v in sorted(per[d].items())))
PY
rm -rf gpurun_out/r05q/*/*/*kernel_trace.csv gpurun_out/r05q/*/*/*agent_info.csv gpurun_out/r05q/*/*/*counter_collection.csv
