"""Measurement: variants of the first pile kernel compared INSIDE one process - the same context, the same allocations, the
variants alternating step by step (option debug_pile_variant; the library built with RALA_HIPCC_FLAGS=-DRALA_PILE_AB carries them).

    python tools/pile_ab.py [workload] [variants, comma separated] [rounds] [steps per round]

Prints per variant the pile chain's time (HIP events around its launches) of every round, their minimum and median."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
variants = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,3").split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ds = Dataset.config(wl)
if os.environ.get("RALA_AB_TORCH_FIRST"):        # (bench.py and the tests have torch's runtime up before the first context)
    import torch
    torch.zeros(1, device="cuda")
ctx = hip.Context(0)
for kv in filter(None, os.environ.get("RALA_AB_OPTIONS", "").split(",")):      # key=value,... (context options)
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.set_reads(ds.read_len)
ctx.set_overlaps(ds.overlaps)
for _ in range(2):
    ctx.initialize()
per = {v: [] for v in variants}
bucket = []
for r in range(rounds):
    order = variants if r % 2 == 0 else variants[::-1]
    for v in order:
        ctx.set_option("debug_pile_variant", v)
        tot = 0.0
        for _ in range(steps):
            ctx.initialize()
            tot += ctx.timings()["pile_ms"]
            bucket.append(ctx.timings()["bucket_ms"])
        per[v].append(tot / steps)
ctx.set_option("debug_pile_variant", 0)
ctx.initialize()
ctx.construct()
print("transitive pairs", ctx.remove_transitive_edges(), "| bucketing (the same kernels in every step) min %.3f median %.3f ms" % (min(bucket), statistics.median(bucket)))
for v in variants:
    x = per[v]
    print("variant %d: pile min %.3f median %.3f | %s" % (v, min(x), statistics.median(x), " ".join("%.3f" % t for t in x)))
