# A/B inside one call: the first pile kernel over all reads (default) against the class-order indirection
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_reads" > gpurun_out/r2_long_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r2_long_pytest.log | tail -3
for i in 1 2 3; do
for mode in default order; do
if [ $mode = order ]; then export RALA_PILE_FORCE_ORDER=1; else unset RALA_PILE_FORCE_ORDER; fi
python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/r2_ab_bench.json
python -c "import json,sys; d=json.load(open('gpurun_out/r2_ab_bench.json')); print(sys.argv[1], '%.3f ms/step  frac %.3f  pile %.3f bucket %.3f' % (d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['pile_ms'], d['stage_ms']['bucket_ms']))" $mode
done
done
unset RALA_PILE_FORCE_ORDER
timeout 600 python tools/long_read_probe.py 2>&1 | grep "^x"
