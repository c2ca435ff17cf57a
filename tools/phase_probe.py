"""Diagnostic: cumulative time of the pile kernel's phases (stop_after sweep) on c2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
ds = Dataset.config(wl)
ctx = hip.Context(0)
ctx.set_reads(ds.read_len); ctx.set_overlaps(ds.overlaps)
prev = 0.0
for k in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(0, 10))) + [99]:
    ctx.set_option("debug_pile_stop_after", k)
    best = 1e9
    for _ in range(3):
        try:
            ctx.initialize()
        except hip.RalaHipError as e:
            if e.code != -4: raise
        best = min(best, ctx.timings()["pile_ms"])
    print("stop_after %2d: pile_ms %8.3f  (+%.3f)" % (k, best, best - prev), flush=True)
    prev = best
