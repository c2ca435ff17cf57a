# A/B on one box: duplicate removal on the side stream (default) vs on the main stream
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); s=d["stage_ms"]; print("%.3f ms/step  dedupe %.3f bucket %.3f pile %.3f  classify %.3f finish %.3f tail %.3f" % (d["ms_per_step"], s["dedupe_ms"], s["bucket_ms"], s["pile_ms"], s["classify_ms"], s["finish_ms"], s["tail_host_ms"]))'
for v in 1 0 1 0 1 0; do
  echo -n "use_side_stream=$v: "; RALA_BENCH_OPTIONS=use_side_stream=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
done
