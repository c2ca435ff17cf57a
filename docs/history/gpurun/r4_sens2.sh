ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_host_api.py tests/test_gpu_unbounded.py tests/test_gpu_fullsize.py -x -q -k "sens or saw or host or repeat" 2>&1 | tail -4
for wl in c3s c5s; do
RALA_HIP_TRACE=1 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/${wl}_line.json 2> $OUT/${wl}_trace.log
grep "rep:\|sens" $OUT/${wl}_trace.log | tail -17
python3 -c "
import json
d=json.loads(open('$OUT/${wl}_line.json').read().strip().splitlines()[-1]); print('$wl', round(d['ms_per_step'],2), d['sensitive_pass'])"
done
