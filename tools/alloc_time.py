"""Measurement (round 6): what the first rala_hip_initialize of a context costs (it allocates everything) with the rows' buffer
from one hipMalloc and as chunks of 1 GB; alternating, fresh contexts.

    python tools/alloc_time.py [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

ds = Dataset.config(sys.argv[1] if len(sys.argv) > 1 else "c3")
for rnd in range(4):
    for mb in (0, 1024):
        ctx = hip.Context(0)
        ctx.set_option("pile_chunk_mb", mb)
        ctx.set_reads(ds.read_len)
        ctx.set_overlaps(ds.overlaps)
        t0 = time.perf_counter()
        ctx.initialize()
        t1 = time.perf_counter()
        ctx.initialize()
        t2 = time.perf_counter()
        print("chunks of %4d MB: first initialize %.1f ms on the host's clock, second %.2f ms (pile chain %.3f ms)" % (mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ctx.timings()["pile_ms"]), flush=True)
        t3 = time.perf_counter()
        ctx.close()
        print("                   close %.1f ms" % ((time.perf_counter() - t3) * 1e3), flush=True)
