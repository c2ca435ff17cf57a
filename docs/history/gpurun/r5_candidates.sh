# round 5: survivor_masks / gather with everything a candidate needs behind its ids in one round trip, against step by step
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
run() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('classify %.3f finish %.3f step %.3f tr %d' % (d['stage_ms']['classify_ms'], d['stage_ms']['finish_ms'], d['ms_per_step'], d['config']['transitive_pairs']))"; }
for round in 1 2; do
for def in "-DRALA_CANDIDATES_STEP_BY_STEP" ""; do
  touch rala_amd/csrc/overlap_kernels.hip
  RALA_HIPCC_FLAGS="$def" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
  echo "[$def] round $round c3: $(run --steps 10 --warmup 2)"
  [ $round = 1 ] && echo "[$def] round $round c5: $(run --workload c5 --steps 4 --warmup 1)"
done
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2
timeout 600 python tests/fuzz_parity.py 60 60000 2>&1 | tail -1
