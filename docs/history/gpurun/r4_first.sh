# round 4, first look: the default line (CPU baseline on C3 itself), c3s (sensitive pass in the step) with its trace and kernel statistics
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
( time python bench.py --steps 20 --warmup 3 > $OUT/r04_c3_bench.json 2> $OUT/bench.log ) 2> $OUT/bench_time.txt
tail -3 $OUT/bench_time.txt
RALA_HIP_TRACE=1 python bench.py --workload c3s --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/c3s_trace.json 2> $OUT/c3s_trace.log
python bench.py --workload c3s --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r04_c3s_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --workload c3s --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/r04_c3s_kernel_stats.csv
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/stats/*/*kernel_trace.csv | head -1) ALL > $OUT/r04_c3s_step_trace.txt
rm -rf $OUT/stats
cd $ROOT
python3 -c "
import json
d=json.load(open('$OUT/r04_c3_bench.json')); print('c3', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['stage_frac']); print(d['stage_ms']); print(d['cpu_baseline']); print(d.get('end_to_end_from_paf'))
d=json.load(open('$OUT/r04_c3s_bench.json')); print('c3s', d['ms_per_step'], d['value']); print(d['stage_ms']); print(d['sensitive_pass'])
"
grep "rep:\|sens" $OUT/c3s_trace.log | tail -40
head -30 $OUT/r04_c3s_kernel_stats.csv
