# round 6, final build: the default bench line from three processes after each other (a process' pile kernel lands between 3.8 and
# 4.25 ms on one box - DESIGN.md section 5), then the L2's hit rate under classify at C3 and C5
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r06/r06_c3_bench_p$i.json 2> gpurun_out/r06/bench_p$i.log
  python -c "
import json
d=json.load(open('gpurun_out/r06/r06_c3_bench_p$i.json'))
print('process $i: %.2f ms = %.2f G overlaps/s, pile chain %.3f ms, frac %.3f, achievable %.3f, stage %.3f, check %s' % (d['ms_per_step'], d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['frac_of_achievable'], d['roofline']['stage_frac'], d['result_check']['ok']))"
done
timeout 900 bash tools/gpurun/r6_classify_l2.sh 2>&1 | tail -6
