# round 5: the sensitive pass with its position-space kernels beside the run-space ones and four looks from the host
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05d
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_unbounded.py tests/test_gpu_host_api.py tests/test_gpu_sharded.py -m gpu -x -q -k "sens or repeat or saw or unbounded or pits_and_hills" 2>&1 | tail -8
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
q --workload c3s --steps 10 --warmup 2 > $OUT/c3s.json
q --workload c5s --steps 4 --warmup 1 > $OUT/c5s.json
for f in c3s c5s; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), d['sensitive_pass']['ms'], {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statss -- python3 $ROOT/bench.py --workload c3s --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/statss.log
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/statss/*/*kernel_trace.csv | head -1) ALL > $OUT/r05_c3s_step_trace.txt
cp $(ls $OUT/statss/*/*kernel_stats.csv | head -1) $OUT/r05_c3s_kernel_stats.csv
rm -rf $OUT/statss
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $ROOT/bench.py --workload c5s --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/stats5.log
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/stats5/*/*kernel_trace.csv | head -1) ALL > $OUT/r05_c5s_step_trace.txt
cp $(ls $OUT/stats5/*/*kernel_stats.csv | head -1) $OUT/r05_c5s_kernel_stats.csv
rm -rf $OUT/stats5
