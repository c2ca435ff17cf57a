# kernel statistics of a short C3 bench run: gpurun_out/r3_kernel_stats.csv (+ the top rows on stdout)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/stats -- python3 $ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e ${RALA_BENCH_ARGS} > $ROOT/gpurun_out/stats_bench.json 2> $ROOT/gpurun_out/stats.log
f=$(ls $ROOT/gpurun_out/stats/*/*kernel_stats.csv | head -1)
cp $f $ROOT/gpurun_out/r3_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/r3_kernel_stats.csv")))
for r in rows[:${1:-40}]:
    print("%-70s calls %5s avg %9.1f us  %5.2f%%" % (r['Name'].replace('rala_hip::','').replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
rm -rf $ROOT/gpurun_out/stats
