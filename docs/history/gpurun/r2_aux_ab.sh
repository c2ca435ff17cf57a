# A/B inside one call: the pile chain's small kernels beside the first one (default) against one stream
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
for mode in aux one; do
if [ $mode = one ]; then export RALA_PILE_NO_AUX=1; else unset RALA_PILE_NO_AUX; fi
python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/r2_ab_bench.json
python -c "import json,sys; d=json.load(open('gpurun_out/r2_ab_bench.json')); print(sys.argv[1], '%.3f ms/step  frac %.3f  pile %.3f bucket %.3f' % (d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['pile_ms'], d['stage_ms']['bucket_ms']))" $mode
done
done
