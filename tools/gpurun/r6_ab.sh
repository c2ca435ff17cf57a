# round 6: alternating A/B of pile-kernel variant builds in one box (RALA_HIPCC_FLAGS): every variant built once, the libraries
# swapped between the runs, R6_ROUNDS (4) alternations of R6_STEPS (20) steps; per variant the pile chain's times, their minimum and
# median.  (One alternation of ten steps moved by +-0.25 ms for the SAME build in round 6's first call: a single pair decides nothing.)
# usage: r6_ab.sh "<flags A>" "<flags B>" ...   ("" = the default build); R6_C5=1 adds one C5 line per variant;
# R6_PARITY=0 skips the parity suites + fuzzer on the default build at the end
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
SO=rala_amd/csrc/librala_hip.so
run() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
i=0
for def in "$@"; do
  i=$((i+1))
  touch rala_amd/csrc/pile_runs_kernel.hip
  RALA_HIPCC_FLAGS="$def" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
  cp $SO /tmp/r6_variant_$i.so
done
: > /tmp/r6_ab.txt
for round in $(seq 1 ${R6_ROUNDS:-4}); do
  i=0
  for def in "$@"; do
    i=$((i+1))
    cp /tmp/r6_variant_$i.so $SO
    line=$(run --steps ${R6_STEPS:-20} --warmup 2)
    echo "[$def] round $round c3: $line"
    echo "$i $line" >> /tmp/r6_ab.txt
    [ $round = 1 ] && [ -n "$R6_C5" ] && echo "[$def] round $round c5: $(run --workload c5 --steps 4 --warmup 1)"
  done
done
python - "$@" <<'PY'
import sys, statistics
names = sys.argv[1:]
per = {}
for l in open("/tmp/r6_ab.txt"):
    f = l.split()
    if len(f) > 4:
        per.setdefault(int(f[0]), []).append((float(f[2]), float(f[4])))
for i, v in sorted(per.items()):
    p = [x[0] for x in v]; s = [x[1] for x in v]
    print("variant %d [%s]: pile min %.3f median %.3f | step min %.3f median %.3f | pile runs %s" % (i, names[i - 1], min(p), statistics.median(p), min(s), statistics.median(s), " ".join("%.3f" % x for x in p)))
PY
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
if [ "${R6_PARITY:-1}" = 1 ]; then
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_wrap.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python tests/fuzz_parity.py ${R6_FUZZ:-60} 2>&1 | tail -1
fi
