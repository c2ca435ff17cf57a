# A/B of two versions of the pile kernel on the same box: tools/_bin/pile_runs_kernel_{old,new}.hip
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); s=d["stage_ms"]; print("%.3f ms/step  pile %.3f  bucket %.3f  classify %.3f finish %.3f tail %.3f" % (d["ms_per_step"], s["pile_ms"], s["bucket_ms"], s["classify_ms"], s["finish_ms"], s["tail_host_ms"]))'
for v in new old new old; do
  cp tools/_bin/pile_runs_kernel_$v.hip rala_amd/csrc/pile_runs_kernel.hip
  python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error
  echo -n "$v: "; python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
done
