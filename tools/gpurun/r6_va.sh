# round 6: does the pile kernel's mode (3.9 or 4.1 ms at C3, per process) follow where its buffers lie?  Addresses + times of several
# processes, with and without torch's runtime initialised first
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
touch rala_amd/csrc/pile_runs_kernel.hip
RALA_HIPCC_FLAGS="-DRALA_PILE_AB $R6_FLAGS" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
for k in $(seq 1 ${R6_PROCS:-6}); do
  echo "plain, process $k: $(RALA_HIP_TRACE_BUFFERS=1 python tools/pile_ab.py c3 "$1" 2 4 2>&1 | grep "variant\|buffers" | sort -u | tr '\n' ' ')"
  echo "torch first, process $k: $(RALA_AB_TORCH_FIRST=1 RALA_HIP_TRACE_BUFFERS=1 python tools/pile_ab.py c3 "$1" 2 4 2>&1 | grep "variant\|buffers" | sort -u | tr '\n' ' ')"
done
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
