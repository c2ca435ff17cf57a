cd $GRAFT_REPO_ROOT
for e in 0 2100 5600; do
RALA_PILE_EXTRA_LDS=$e python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/occ.json
python -c "import json,sys; d=json.load(open('gpurun_out/occ.json')); print('extra lds', sys.argv[1], 'pile %.3f' % (d['stage_ms']['pile_ms']))" $e
done
