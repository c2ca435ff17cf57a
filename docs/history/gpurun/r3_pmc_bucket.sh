# round 3: memory-side counters of bucket_fixed_kernel at C3 - bytes fetched / written against the
# 2.03 GB it moves algorithmically, atomics and write requests per overlap.  One rocprofv3 --pmc pass per group.
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_bucket
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o "TCC_[A-Z_0-9]*" $OUT/avail.txt | sort -u | tr '\n' ' ' | cut -c1-4000; echo
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass atom TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass req TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_HIT_sum
pass miss TCC_MISS_sum TCC_WRITEBACK_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("gpurun_out/pmc_bucket/*/*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "bucket_fixed_kernel" in k:
            acc[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    for d in sorted(acc)[-1:]:
        out.update(acc[d])
n_ovl = 50858245
print(json.dumps(out, indent=1))
alg = 40.0 * n_ovl
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    fetched = 2.0 * out["FETCH_SIZE"] * 1024.0      # gfx950: FETCH_SIZE counts half of wide streaming reads (guide); KB -> bytes
    written = out["WRITE_SIZE"] * 1024.0
    print("algorithmic %.2f GB; fetched (x2 corrected) %.2f GB; written %.2f GB; written / the 16 B per overlap of bounds: %.2fx" % (
        alg / 1e9, fetched / 1e9, written / 1e9, written / (16.0 * n_ovl)))
for k in ("TCC_ATOMIC_sum", "TCC_EA0_ATOMIC_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_REQ_sum", "TCC_WRITE_sum", "TCC_READ_sum"):
    if k in out:
        print("%s per overlap: %.2f" % (k, out[k] / n_ovl))
json.dump(out, open("gpurun_out/pmc_bucket/summary.json", "w"), indent=1)
PY
rm -rf $OUT/*/*/*kernel_trace.csv $OUT/*/*/*agent_info.csv
