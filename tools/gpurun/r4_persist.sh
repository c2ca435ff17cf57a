# pile kernel: persistent wavefronts with the next read's events requested one read ahead (RALA_PILE_PERSIST2=<grid>)
# against one workgroup per read; parity first (the C2 / C3 digests and the parity suite with the persistent kernel).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RALA_PILE_PERSIST2=7168 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_unbounded.py -m gpu -x -q 2>&1 | tail -5
RALA_PILE_PERSIST2=7168 timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c2 or c3" 2>&1 | tail -5
run() { python bench.py --no-cpu-baseline --no-e2e --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pile %.3f step %.3f overflow %d tr %d' % (d['stage_ms']['pile_ms'], d['ms_per_step'], d['stage_ms']['pile_overflow_reads'], d['config']['transitive_pairs']))"; }
for k in 1 2; do
  echo "one per read   : $(run)"
  echo "persist 7168   : $(RALA_PILE_PERSIST2=7168 run)"
  echo "persist 6144   : $(RALA_PILE_PERSIST2=6144 run)"
  echo "persist 14336  : $(RALA_PILE_PERSIST2=14336 run)"
  echo "persist 5120   : $(RALA_PILE_PERSIST2=5120 run)"
done
