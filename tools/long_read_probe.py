"""Diagnostic: the pile chain on a data set whose reads are all longer than 16384 bases (the C2 set with every
coordinate doubled): they skip the first kernel of the chain and run in the any-length cap-512 kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rala_amd import hip
from rala_amd.synth import Dataset, Overlaps, FIELDS

ds = Dataset.config("c2")
for factor in ([int(a) for a in sys.argv[1:]] or [1, 2, 5, 8]):
    ov = ds.overlaps
    kw = {f: getattr(ov, f) for f in FIELDS}
    for f in ("a_begin", "a_end", "b_begin", "b_end", "length"):
        kw[f] = kw[f] * factor
    o2 = Overlaps(strand=ov.strand, **kw)
    ctx = hip.Context(0)
    ctx.set_reads((ds.read_len * factor).astype(np.uint32))
    ctx.set_overlaps(o2)
    best = 1e9
    for _ in range(5):
        ctx.initialize()
        tm = ctx.timings()
        best = min(best, tm["pile_ms"])
    gb = (16.0 * len(o2) + 2.0 * float((ds.read_len * factor).sum()) + 40.0 * ds.n_reads) / 1e9
    print("x%d: %d reads, mean length %.0f, pile chain %.3f ms = %.0f GB/s (%.2f of the HBM peak), overflow past cap 512: %d"
          % (factor, ds.n_reads, float((ds.read_len * factor).mean()), best, gb / best * 1e3, gb / best * 1e3 / 8000.0, tm["pile_overflow_reads"]))
