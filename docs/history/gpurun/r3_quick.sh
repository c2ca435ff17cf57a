# parity subset + a bench line: the loop of round 3 (everything also in gpurun_out/q.log)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_wrap.py tests/test_gpu_sharded.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/q.log
python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.3f  frac %.3f stage_frac %.3f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_frac'])); print({k: round(v,3) for k,v in d['stage_ms'].items()})" >> gpurun_out/q.log
if [ -n "$1" ]; then
  bash tools/gpurun/r3_trace.sh > /dev/null
  head -1 gpurun_out/step_trace.txt >> gpurun_out/q.log
fi
tail -12 gpurun_out/q.log
