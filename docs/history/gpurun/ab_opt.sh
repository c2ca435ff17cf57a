# A/B of a context option on one box: bash tools/gpurun/ab_opt.sh <option>   (values 1 and 0, three times each)
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); s=d["stage_ms"]; print("%.3f ms/step  bucket %.3f pile %.3f classify %.3f death %.3f (%d rounds) finish %.3f tail %.3f tr %.3f" % (d["ms_per_step"], s["bucket_ms"], s["pile_ms"], s["classify_ms"], s["death_ms"], s["death_rounds"], s["finish_ms"], s["tail_host_ms"], s["tr_ms"]))'
for v in 1 0 1 0 1 0; do
  echo -n "$1=$v: "; RALA_BENCH_OPTIONS=$1=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
done
