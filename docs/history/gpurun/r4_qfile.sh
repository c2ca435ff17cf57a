# EXPERIMENT, not in the tree: apply tools/experiments/r4_query_from_file.patch first (git apply).  The query side of the bound
# rows straight from the file's columns against rows completed by query_side_kernel (RALA_QUERY_ROWS=1),
# one box; parity first.  Result (round 4): parity green; bucketing 1.28 -> 1.01 ms, pile kernel 4.08 -> 4.55 ms (4.24 with the
# code in place and the rows complete): step 7.81 against 7.77 ms at C3, 36.4 against 37.0 at C5 - not kept.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_unbounded.py tests/test_gpu_wrap.py tests/test_gpu_edges.py tests/test_gpu_host_api.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c2 or c3" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-e2e --steps $2 --warmup 2 $1 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f stage %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_frac'], d['config']['transitive_pairs']))"; }
for k in 1 2; do
  echo "c3 from the file : $(run '' 10)"
  echo "c3 rows          : $(RALA_QUERY_ROWS=1 run '' 10)"
done
echo "c5 from the file : $(run '--workload c5' 4)"
echo "c5 rows          : $(RALA_QUERY_ROWS=1 run '--workload c5' 4)"
echo "c3s from the file : $(run '--workload c3s' 6)"
echo "c3s rows          : $(RALA_QUERY_ROWS=1 run '--workload c3s' 6)"
