# A/B of context options on the C3 bench line: bash tools/gpurun/r2_opts.sh "use_side_stream=0" "use_round_batches=0" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for o in "" "$@" ""; do
RALA_BENCH_OPTIONS="$o" python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/opts.json
python -c "import json,sys; d=json.load(open('gpurun_out/opts.json')); s=d['stage_ms']; print('%-24s %.3f ms/step  dedupe %.2f bucket %.2f pile %.2f' % (sys.argv[1] or 'default', d['ms_per_step'], s['dedupe_ms'], s['bucket_ms'], s['pile_ms']))" "$o"
done
