# round 6: ev_off in bound pairs (the partitioned bucketing): parity suites, both fuzzers, the box's host memory
cd $GRAFT_REPO_ROOT
free -g | head -2; nproc
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_wrap.py tests/test_gpu_rows.py tests/test_gpu_sharded.py tests/test_gpu_unbounded.py tests/test_gpu_layout.py tests/test_gpu_host_api.py tests/test_gpu_ingest.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 900 python tests/fuzz_parity.py 60 2>&1 | tail -1
timeout 900 python tests/fuzz_sharded.py 30 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-e2e 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench: %.2f ms, pile %.3f, bucket %.3f, frac %.3f, check %s' % (d['ms_per_step'], d['stage_ms']['pile_ms'], d['stage_ms']['bucket_ms'], d['roofline']['frac'], d['result_check']['ok']))"
