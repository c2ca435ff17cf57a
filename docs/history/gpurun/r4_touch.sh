# EXPERIMENT, not in the tree (round 4; no gain: 4.11 - 4.21 ms at every distance against 4.05 - 4.12).  First pile kernel: a
# wavefront also asks for the first kilobyte of the events of the read RALA_PILE_TOUCH reads ahead (a read
# a wavefront of the same XCD starts on a few microseconds later), one box
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-e2e --steps 12 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket %.3f pile %.3f step %.3f frac %.3f tr %d' % (d['stage_ms']['bucket_ms'], d['stage_ms']['pile_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['transitive_pairs']))"; }
for k in 1 2; do
  echo "no touch   : $(run)"
  for d in 256 512 1024 2048 4096 8192; do echo "touch $d : $(RALA_PILE_TOUCH=$d run)"; done
done
