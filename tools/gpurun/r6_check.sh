# round 6: parity of the current build (pile / bucketing paths, every row, both fuzzers), then bench lines
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_wrap.py tests/test_gpu_rows.py tests/test_gpu_unbounded.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python tests/fuzz_parity.py ${R6_FUZZ:-60} 2>&1 | tail -1
timeout 600 python tests/fuzz_sharded.py ${R6_FUZZ_SHARDED:-20} 2>&1 | tail -1
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f frac %.3f stage_frac %.3f tr %d check %s' % (d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_frac'], d['config']['transitive_pairs'], (d.get('result_check') or {}).get('ok')))"; }
for k in 1 2 3; do echo "c3: $(q --steps 20 --warmup 3)"; done
echo "c5: $(q --workload c5 --steps 4 --warmup 1)"
echo "c3s: $(q --workload c3s --steps 6 --warmup 2)"
