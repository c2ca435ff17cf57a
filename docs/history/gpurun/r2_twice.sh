# the C3 bench line four times in one call (boxes differ by up to 10 %: compare inside a call only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3 4; do
RALA_BENCH_OPTIONS="$1" python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/twice.json
python -c "import json; d=json.load(open('gpurun_out/twice.json')); s=d['stage_ms']; print('%.3f ms/step  bucket %.2f pile %.3f' % (d['ms_per_step'], s['bucket_ms'], s['pile_ms']))"
done
