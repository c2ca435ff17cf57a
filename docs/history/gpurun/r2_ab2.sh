# parity incl. the sharded runs + the full-size variants, then a C3 bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrap.py tests/test_gpu_golden.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "not c5" 2>&1 | tail -3
for i in 1 2; do
python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/r2_ab_bench.json
python -c "import json; d=json.load(open('gpurun_out/r2_ab_bench.json')); print('%.3f ms/step  %.3f G ovl/s  frac %.3f' % (d['ms_per_step'], d['value']/1e9, d['roofline']['frac'])); print({k: round(v, 3) for k, v in d['stage_ms'].items()})"
done
