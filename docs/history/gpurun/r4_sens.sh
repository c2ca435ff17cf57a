# the sensitive pass: parity tests, then c3s with the stage trace
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_host_api.py tests/test_gpu_unbounded.py -x -q -k "sens or saw or unbounded or host or repeat" 2>&1 | tail -5
RALA_HIP_TRACE=1 python bench.py --workload c3s --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/c3s_trace.json 2> $OUT/c3s_trace.log
grep "rep:\|sens" $OUT/c3s_trace.log | tail -14
python bench.py --workload c3s --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" > $OUT/r04_c3s_bench.json
python3 -c "
import json
d=json.load(open('$OUT/r04_c3s_bench.json')); print('c3s', d['ms_per_step'], d['value']); print(d['stage_ms']); print(d['sensitive_pass'])
"
