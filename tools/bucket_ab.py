"""Measurement: the bucketing's kernels with their units handed out in XCD ranges (option debug_bucket_xcd, a mask: 1 first scatter,
2 second, 4 rows, 8 query side) against every eighth, inside one process, alternating; the events must come out the same
(multisets per read: checked through the piles' digests of a full step)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
masks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,15,1,2,4,8").split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ds = Dataset.config(wl)
ctx = hip.Context(0)
ctx.set_reads(ds.read_len)
ctx.set_overlaps(ds.overlaps)
ctx.initialize()
ref = None
per = {m: [] for m in masks}
pile = {m: [] for m in masks}
for r in range(rounds):
    for m in (masks if r % 2 == 0 else masks[::-1]):
        ctx.set_option("debug_bucket_xcd", m)
        t = p = 0.0
        for _ in range(4):
            ctx.initialize()
            t += ctx.timings()["bucket_ms"]; p += ctx.timings()["pile_ms"]
        per[m].append(t / 4); pile[m].append(p / 4)
        if r == 0:
            fnv, inside, _ = ctx.pile_row_digests()
            key = (int(fnv.sum(dtype="uint64")), int(inside.sum(dtype="uint64")))
            ref = ref or key
            assert key == ref, (m, key, ref)
for m in masks:
    print("mask %2d: bucketing min %.3f median %.3f | pile median %.3f | %s" % (m, min(per[m]), statistics.median(per[m]), statistics.median(pile[m]), " ".join("%.3f" % t for t in per[m])))
