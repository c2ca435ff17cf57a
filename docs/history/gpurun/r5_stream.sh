# round 5: classify / survivor masks with ordinary loads of the columns instead of non-temporal ones, two alternations
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('classify %.3f finish %.3f step %.3f tr %d' % (d['stage_ms']['classify_ms'], d['stage_ms']['finish_ms'], d['ms_per_step'], d['config']['transitive_pairs']))"; }
for round in 1 2; do
for def in "" "-DRALA_STREAM_PLAIN"; do
  touch rala_amd/csrc/overlap_kernels.hip
  RALA_HIPCC_FLAGS="$def" python -c "from rala_amd import build; build.build_hip()" 2>&1 | tail -2
  echo "[$def] round $round c3: $(run)"
  [ $round = 1 ] && echo "[$def] round $round c5: $(run --workload c5 --steps 4 --warmup 1)"
done
done
touch rala_amd/csrc/overlap_kernels.hip
python -c "from rala_amd import build; build.build_hip()"
