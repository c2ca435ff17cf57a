# round 5: the sensitive pass with the cap-2048 run-space class and the short layout of the cap-1024 kernels: the whole GPU
# suite, both fuzzers, C3 / C5 with -s, the C5 trace
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05s
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
timeout 900 python tests/fuzz_parity.py 100 2>&1 | tail -1
timeout 900 python tests/fuzz_sharded.py 40 2>&1 | tail -1
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
q --steps 10 --warmup 2 > $OUT/c3.json
q --workload c3s --steps 6 --warmup 2 > $OUT/c3s.json
q --workload c5s --steps 3 --warmup 1 > $OUT/c5s.json
for f in c3 c3s c5s; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), round(d['roofline']['frac'],3), {k: round(v,2) for k,v in d['stage_ms'].items() if isinstance(v,float) and v})"; done
RALA_HIP_TRACE=1 python bench.py --workload c5s --no-cpu-baseline --no-e2e --steps 1 --warmup 0 2>&1 | grep "sens pass" | head -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $ROOT/bench.py --workload c5s --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/tr.log
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) ALL > $OUT/c5s_step_trace.txt
rm -rf $OUT/tr
grep -n "sens_records" -A70 $OUT/c5s_step_trace.txt | grep "  at " | cut -c1-130 | head -70
