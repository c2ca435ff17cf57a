# list kernels with one add per workgroup (targets / members of the sensitive pass, event-dense reads): parity of the -s paths, then c3s / c5s / c5
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "sens or c2 or c3" 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --no-e2e --steps $2 --warmup 1 $1 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f frac %.3f sens %s tr %d' % (d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d.get('sensitive_pass',{}).get('ms'), d['config']['transitive_pairs']))"; }
echo "c3s : $(run '--workload c3s' 8)"
echo "c5s : $(run '--workload c5s' 4)"
echo "c5  : $(run '--workload c5' 4)"
