# round 5: the expansion's two value look-ups without their zero extension (two vector instructions per 16-byte store),
# against the build that extends them; two alternations at C3, one at C5, then parity on the final build
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
run() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for round in 1 2; do
for def in "-DRALA_EXPAND_EXTEND" ""; do
  touch rala_amd/csrc/pile_runs_kernel.hip
  RALA_HIPCC_FLAGS="$def" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
  echo "[$def] round $round c3: $(run --steps 10 --warmup 2)"
  [ $round = 1 ] && echo "[$def] round $round c5: $(run --workload c5 --steps 4 --warmup 1)"
done
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_wrap.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python tests/fuzz_parity.py 60 2>&1 | tail -1
