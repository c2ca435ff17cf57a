# (needs docs/history/experiments/r6_pile_kernel_variants.patch applied - the probe is no product code - and the line that prints
# d_small[12 .. 13] behind rala_hip_initialize, which went with it)
# round 6: does workgroup i run on XCD i % 8 - always, or only in a box's fast state?  The first pile kernel counts the workgroups that
# do not (variant bit 22), several processes, the kernel's time beside the count
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
touch rala_amd/csrc/pile_runs_kernel.hip
RALA_HIPCC_FLAGS="-DRALA_PILE_AB '-DRALA_PILE_AB_CASES=X(4194304) X(4202496)'" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
for k in $(seq 1 ${R6_PROCS:-6}); do
  echo "process $k:"
  RALA_HIP_TRACE_PROBE=1 python tools/pile_ab.py c3 4194304,4202496 2 3 2>&1 | grep "probe\|variant" | awk '{print}' | sort | uniq -c | sort -rn | head -8
done
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
