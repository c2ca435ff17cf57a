ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_c3 -- python3 $ROOT/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_c3_bench.json 2> $ROOT/gpurun_out/prof_c3.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_fetch -- python3 $ROOT/bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_write -- python3 $ROOT/bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/pmc_write.log
cd $ROOT
find gpurun_out/prof_c3 gpurun_out/pmc_fetch gpurun_out/pmc_write -type f | head -20
rm -f gpurun_out/prof_c3/*/*kernel_trace.csv
for d in pmc_fetch pmc_write; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); echo $f; head -3 $f; python3 - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-60:]
    tot[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for (k, c), v in sorted(tot.items(), key=lambda x: -x[1])[:12]:
    print("%-62s %-12s calls %4d  sum %.1f" % (k, c, cnt[(k, c)], v))
PY
rm -f $f; done
cat gpurun_out/prof_c3_bench.json | cut -c1-300
