# round 2: parity suite (without the opt-in sizes) + a C3 bench line, one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "[pytest: $(( $(date +%s) - T0 )) s]"
python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/r2_quick_bench.json
python -c "import json; d=json.load(open('gpurun_out/r2_quick_bench.json')); print('%.3f ms/step  %.3f G ovl/s  frac %.3f' % (d['ms_per_step'], d['value']/1e9, d['roofline']['frac'])); print(d['stage_ms'])"
