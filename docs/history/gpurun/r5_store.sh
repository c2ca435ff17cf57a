# round 5: cache-policy bits of the pile kernel's row stores (plain / sc1 / sc0 sc1 / nt), two alternations on one box;
# + the windowed ingest test
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05g
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q -k "windows or tokeniser_matches" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for round in 1 2; do
for mod in "" "sc1" "sc0 sc1" "nt"; do
  touch rala_amd/csrc/pile_runs_kernel.hip
  RALA_HIPCC_FLAGS="-DRALA_ROW_STORE_MOD=\"\\\"$mod\\\"\"" python -c "from rala_amd import build; build.build_hip()" 2>&1 | tail -2
  echo "row stores [$mod] round $round: $(run)"
done
done
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()"
