# round 5: what the driver runs at round end - smoke(), the GPU suite, the default bench line
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
( time python bench.py ) > gpurun_out/r5_last_bench.json 2> gpurun_out/r5_last_bench.err; tail -4 gpurun_out/r5_last_bench.err
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r5_last_bench.json') if l.startswith('{')][0]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['from_pinned_host'].get('value'))"
