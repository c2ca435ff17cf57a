# reads per workgroup in the first pile kernel: A/B inside one call
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for w in 1 2 4; do
  RALA_PILE_WAVES=$w python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves', $w, 'ms', round(d['ms_per_step'],3), 'pile', round(d['stage_ms']['pile_ms'],3), 'pairs', d['config']['transitive_pairs'])"
done
done
RALA_PILE_WAVES=4 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -x -q -k "golden or full_path" 2>&1 | tail -3
