# the tail's time over several bench invocations (is it steady?)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/r2_ab_bench.json
python -c "import json,sys; d=json.load(open('gpurun_out/r2_ab_bench.json')); s=d['stage_ms']; print('%.3f ms/step  pile %.3f bucket %.3f death %.3f finish %.3f tail %.3f tr %.3f' % (d['ms_per_step'], s['pile_ms'], s['bucket_ms'], s['death_ms'], s['finish_ms'], s['tail_host_ms'], s['tr_ms']))"
done
