# round 5: the sensitive pass's run-space kernels with everything of a read behind its id in one round trip, against step by step
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
run() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pass %.3f step %.3f tr %d' % (d['stage_ms']['repeats_ms'], d['ms_per_step'], d['config']['transitive_pairs']))"; }
for round in 1 2; do
for def in "-DRALA_SENS_START_STEP_BY_STEP" ""; do
  touch rala_amd/csrc/pile_runs_kernel.hip
  RALA_HIPCC_FLAGS="$def" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
  echo "[$def] round $round c3s: $(run --workload c3s --steps 6 --warmup 2)"
  echo "[$def] round $round c5s: $(run --workload c5s --steps 3 --warmup 1)"
done
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -m gpu -x -q -k "sensitive or sens" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2
timeout 600 python tests/fuzz_parity.py 60 90000 2>&1 | tail -1
