# round 6: what the driver runs at round end - smoke(), the GPU suite, the default bench line - plus both fuzzers, on the final build
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 900 python tests/fuzz_parity.py 100 2>&1 | tail -1
timeout 900 python tests/fuzz_sharded.py 40 2>&1 | tail -1
python bench.py > gpurun_out/r6_last_bench.json 2> gpurun_out/r6_last_bench.log; tail -2 gpurun_out/r6_last_bench.log
python -c "
import json
d=json.load(open('gpurun_out/r6_last_bench.json'))
print('bench: %.2f ms = %.2f G overlaps/s, frac %.3f, achievable %.3f, stage %.3f, traffic %.3g, check %s' % (d['ms_per_step'], d['value']/1e9, d['roofline']['frac'], d['roofline']['frac_of_achievable'], d['roofline']['stage_frac'], d['roofline']['traffic'] or 0, d['result_check']['ok']))"
