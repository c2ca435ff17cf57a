cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RALA_HIP_TRACE=1 python bench.py --workload c3 --steps 1 --warmup 2 --no-cpu-baseline 2>&1 | grep "trace" | tail -24
python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'])"
