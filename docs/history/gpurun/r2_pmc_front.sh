# round 2: where does the pile kernel's time go?  Issue / fetch / unit-busy counters of the product
# instantiation (c2, 100 k reads = 100 k wavefronts), several --pmc passes, values per read.
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $ROOT/gpurun_out/pmc2/avail.txt 2>&1
grep -o "SQC\?_[A-Z_0-9]*" $ROOT/gpurun_out/pmc2/avail.txt | sort -u | tr '\n' ' ' | cut -c1-6000
echo
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc2/$name -- python3 $ROOT/tools/pile_once.py c2 99,99 > $ROOT/gpurun_out/pmc2/$name.log 2>&1 || tail -3 $ROOT/gpurun_out/pmc2/$name.log
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH
pass c SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS
pass d SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SALU
pass e SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for name in "abcde":
    for f in glob.glob("gpurun_out/pmc2/%s/*/*counter_collection.csv" % name):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "pile_runs_kernel" in k and "512" in k:
                per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        for d in sorted(per)[-1:]:
            print("pass", name, " ".join("%s=%.0f" % (n[3:] if n.startswith("SQ_") else n, v / 1e5) for n, v in sorted(per[d].items())))
PY
rm -rf gpurun_out/pmc2/*/*/*kernel_trace.csv gpurun_out/pmc2/*/*/*agent_info.csv
tail -2 gpurun_out/pmc2/a.log
