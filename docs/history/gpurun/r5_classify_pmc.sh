# round 5: SQ counters of classify_kernel at C3 (per overlap): is it the memory or the vector unit?
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass c SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for name in "abc":
    for f in glob.glob("gpurun_out/r05c/%s/*/*counter_collection.csv" % name):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            for tag in ("classify_kernel", "survivor_masks_kernel", "final_kernel", "l1_scatter_kernel"):
                if tag in k:
                    per[tag][row["Counter_Name"]] += float(row["Counter_Value"])
        for tag, c in per.items():
            print(name, tag, " ".join("%s=%.3g" % (n.replace("SQ_", ""), v) for n, v in sorted(c.items())))
PY
rm -rf gpurun_out/r05c/*/
