# round 6: variants of the first pile kernel inside ONE process (tools/pile_ab.py), R6_PROCS (3) processes after each other - what
# differs between processes (where the allocations land) is then visible beside what differs between the variants
# usage: r6_ab_inproc.sh "<variants, comma separated>" [workload]
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
touch rala_amd/csrc/pile_runs_kernel.hip
RALA_HIPCC_FLAGS="-DRALA_PILE_AB $R6_FLAGS" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
for k in $(seq 1 ${R6_PROCS:-3}); do
  echo "process $k:"
  python tools/pile_ab.py ${2:-c3} "$1" ${R6_ROUNDS:-6} ${R6_STEPS:-5} 2>&1 | tail -$(( $(echo "$1" | tr ',' '\n' | wc -l) + 1 ))
done
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
