cd $GRAFT_REPO_ROOT
for p in 0 1 2 3; do
  RALA_MASKS_NO_LDS=1 RALA_MASKS_PROBE=$p RALA_BENCH_ARGS="" bash tools/gpurun/r3_stats.sh 20 2>/dev/null | grep -i "survivor_masks" | sed "s/^/probe=$p /"
done
