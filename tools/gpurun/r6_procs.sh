# round 6: the spread of the pile kernel's time BETWEEN processes (same box, same binary), and whether it follows how the kernels that
# run beside the main one are queued: the default, eight hardware queues, everything on one stream
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
touch rala_amd/csrc/pile_runs_kernel.hip
RALA_HIPCC_FLAGS="-DRALA_PILE_AB $R6_FLAGS" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
for k in $(seq 1 ${R6_PROCS:-5}); do
  echo "default, process $k: $(python tools/pile_ab.py c3 "$1" 3 4 2>&1 | grep "variant\|bucketing" | tr '\n' ' ')"
  [ -n "$R6_ALL_MODES" ] && echo "8 queues, process $k: $(GPU_MAX_HW_QUEUES=8 python tools/pile_ab.py c3 "$1" 3 4 2>&1 | grep "variant\|bucketing" | tr '\n' ' ')"
  [ -n "$R6_ALL_MODES" ] && echo "one stream, process $k: $(RALA_AB_OPTIONS=use_side_stream=0 python tools/pile_ab.py c3 "$1" 3 4 2>&1 | grep "variant\|bucketing" | tr '\n' ' ')"
done
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
