# round 5: classify with the second half's columns requested while the first half's records are on their way (three dependent
# round trips per wavefront instead of four, six wavefronts per SIMD instead of seven), two alternations
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
run() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('classify %.3f finish %.3f step %.3f tr %d' % (d['stage_ms']['classify_ms'], d['stage_ms']['finish_ms'], d['ms_per_step'], d['config']['transitive_pairs']))"; }
for round in 1 2; do
for def in "" "-DRALA_CLASSIFY_TWO_TRIPS"; do
  touch rala_amd/csrc/overlap_kernels.hip
  RALA_HIPCC_FLAGS="$def" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
  echo "[$def] round $round c3: $(run --steps 10 --warmup 2)"
  [ $round = 1 ] && echo "[$def] round $round c5: $(run --workload c5 --steps 4 --warmup 1)"
done
done
touch rala_amd/csrc/overlap_kernels.hip
python -c "from rala_amd import build; build.build_hip()"
