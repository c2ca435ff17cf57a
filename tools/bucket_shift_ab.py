"""Measurement: the partitioned bucketing with first-level partitions of 4096 / 8192 / 16384 reads (option debug_part_shift; 0 = the
rule: the smallest size that makes at most 256 partitions), alternating inside one process; the piles must come out the same.

    python tools/bucket_shift_ab.py [workload] [shifts, comma separated] [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
shifts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "12,0,13,14").split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ds = Dataset.config(wl)
ctx = hip.Context(0)
ctx.set_reads(ds.read_len)
ctx.set_overlaps(ds.overlaps)
ctx.initialize()
ref = None
per = {m: [] for m in shifts}
for r in range(rounds):
    for m in (shifts if r % 2 == 0 else shifts[::-1]):
        ctx.set_option("debug_part_shift", m)
        t = 0.0
        for _ in range(3):
            ctx.initialize()
            t += ctx.timings()["bucket_ms"]
        per[m].append(t / 3)
        if r == 0:
            fnv, inside, _ = ctx.pile_row_digests()
            key = (int(fnv.sum(dtype="uint64")), int(inside.sum(dtype="uint64")))
            ref = ref or key
            assert key == ref, (m, key, ref)
ctx.set_option("debug_part_shift", 0)
for m in shifts:
    print("shift %2d: bucketing min %.3f median %.3f ms | %s" % (m, min(per[m]), statistics.median(per[m]), " ".join("%.3f" % t for t in per[m])))
