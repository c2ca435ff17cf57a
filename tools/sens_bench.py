"""Time of Graph::construct with a sensitive overlap set (-s) against the plain construct, at a
synthetic configuration: python tools/sens_bench.py [c2|c3|c5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rala_amd import hip
from rala_amd.synth import Dataset

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
ds = Dataset.config(wl)
ctx = hip.Context(0)
ctx.set_reads(ds.read_len)
ctx.set_overlaps(ds.overlaps)


def timed(fn):
    t0 = time.perf_counter()
    fn()
    return 1e3 * (time.perf_counter() - t0)


ctx.initialize(); ctx.construct(); ctx.remove_transitive_edges()        # warm-up
t_init = timed(ctx.initialize)
t_plain = timed(ctx.construct)
n_plain = ctx.remove_transitive_edges()
# the sensitive set is derived from the piles as the chimera stage leaves them
p = ctx.piles()
t0 = time.perf_counter()
sens = ds.sensitive(p["alive"], p["begin"], p["end"])
t_gen = 1e3 * (time.perf_counter() - t0)
ctx.initialize()
t_sens_cold = timed(lambda: ctx.construct(sens))
ctx.initialize()
t_sens = timed(lambda: ctx.construct(sens))                               # buffers exist now
n_sens = ctx.remove_transitive_edges()
offs, pairs, flags = ctx.intervals(2)
print("%s: %d overlaps, %d sensitive overlaps (generated in %.0f ms); initialize %.1f ms, construct %.1f ms, "
      "construct(-s) %.1f ms (first call %.1f ms); repeat hills %d; transitive pairs %d -> %d" % (
          wl, len(ds.overlaps), len(sens), t_gen, t_init, t_plain, t_sens, t_sens_cold, len(pairs), n_plain, n_sens))
