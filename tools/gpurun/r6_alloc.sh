# round 6: the pile kernel's spread between contexts of ONE process (tools/pile_alloc_probe.py), two processes
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
touch rala_amd/csrc/pile_runs_kernel.hip
RALA_HIPCC_FLAGS="-DRALA_PILE_AB $R6_FLAGS" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
for k in 1 2; do echo "process $k:"; RALA_HIP_TRACE_BUFFERS=1 python tools/pile_alloc_probe.py c3 "${1:-3,8192,1024}" 4 4 2>&1 | grep -v "^\[trace\]" | tail -5; done
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
