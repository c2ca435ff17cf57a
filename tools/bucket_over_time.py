"""Diagnostic: the bucketing stage's time over many consecutive full steps of one context (does a process
drift between the 2.4 ms and the 3.3 ms behaviour?).  python tools/bucket_over_time.py [workload] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

ds = Dataset.config(sys.argv[1] if len(sys.argv) > 1 else "c3")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = hip.Context(0)
ctx.set_reads(ds.read_len)
ctx.set_overlaps(ds.overlaps)
t0 = time.time()
row = []
for k in range(steps):
    ctx.initialize()
    b = ctx.timings()["bucket_ms"]
    ctx.construct()
    ctx.remove_transitive_edges()
    row.append(b)
    if len(row) == 20:
        print("t = %5.1f s  steps %3d - %3d: bucket_ms min %.2f  median %.2f  max %.2f" % (
            time.time() - t0, k - 19, k, min(row), sorted(row)[10], max(row)), flush=True)
        row = []
