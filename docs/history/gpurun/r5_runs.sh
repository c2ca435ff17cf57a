# round 5: the query side copied from the queries' runs inside the bucketing's last kernel, against query_side_kernel
# (RALA_QUERY_SIDE_APART=1): two alternations at C3, C5, then the suite and the fuzzers
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05r
mkdir -p $OUT
cd $ROOT
q() { python bench.py --no-cpu-baseline --no-e2e "$@" 2>$OUT/err.log | grep '^{'; }
for k in 1 2; do
q --steps 10 --warmup 2 > $OUT/runs_$k.json
RALA_QUERY_SIDE_APART=1 q --steps 10 --warmup 2 > $OUT/apart_$k.json
done
q --workload c5 --steps 4 --warmup 1 > $OUT/c5_runs.json
RALA_QUERY_SIDE_APART=1 q --workload c5 --steps 4 --warmup 1 > $OUT/c5_apart.json
for f in runs_1 apart_1 runs_2 apart_2 c5_runs c5_apart; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', round(d['ms_per_step'],2), d['config'].get('transitive_pairs'), round(d['roofline']['frac'],3), round(d['roofline']['stage_frac'],3), {k: round(v,3) for k,v in d['stage_ms'].items() if isinstance(v,float) and v and k in ('bucket_ms','pile_ms','total_ms')})"; done
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 900 python tests/fuzz_parity.py 60 130000 2>&1 | tail -1
