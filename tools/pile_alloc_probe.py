"""Measurement (round 6): is the run-to-run spread of the pile kernel a property of WHERE a context's buffers land?  Several contexts
after each other inside one process, each with the same data and the same variants; optionally a large block allocated and freed
(or kept) in between, to move the next context's allocations elsewhere.

    python tools/pile_alloc_probe.py [workload] [variants] [contexts] [steps]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rala_amd import hip
from rala_amd.synth import Dataset

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
variants = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3,8192").split(",")]
contexts = int(sys.argv[3]) if len(sys.argv) > 3 else 4
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ds = Dataset.config(wl)
torch.zeros(1, device="cuda")
keep = []
for c in range(contexts):
    free, total = torch.cuda.mem_get_info()
    ctx = hip.Context(0)
    import time as _t
    _t0 = _t.perf_counter()
    ctx.set_reads(ds.read_len)
    alloc_ms = (_t.perf_counter() - _t0) * 1e3
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    out = []
    for rnd in range(3):
        for v in variants:
            ctx.set_option("debug_pile_variant", v)
            t = 0.0
            for _ in range(steps):
                ctx.initialize()
                t += ctx.timings()["pile_ms"]
            out.append((v, t / steps))
    per = {v: [t for (w, t) in out if w == v] for v in variants}
    print("context %d (free before: %.1f GB, set_reads %.0f ms): %s" % (c, free / 1e9, alloc_ms, "  ".join("var %d: %s" % (v, " ".join("%.3f" % t for t in per[v])) for v in variants)), flush=True)
    ctx.close()
    if c % 2 == 0:
        # a block that stays: the next context's buffers land behind it
        keep.append(torch.empty(int(7e9) * (c + 1), dtype=torch.uint8, device="cuda"))
