"""Diagnostic: the bucketing stage runs in 2.3 ms or in 3.3 ms (C3) depending on the process.  Is it the
placement of the context's buffers?  Several contexts in one process, each measured a few times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

ds = Dataset.config(sys.argv[1] if len(sys.argv) > 1 else "c3")
keep = []
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    ctx = hip.Context(0)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    t = []
    for i in range(12):
        if i == 1:
            os.environ.pop("RALA_HIP_TRACE", None)
        ctx.initialize()
        t.append(ctx.timings()["bucket_ms"])
    if os.environ.get("RALA_PROBE_ADDRESSES"):
        os.environ["RALA_HIP_TRACE"] = "1"          # the next context's first call prints its buffer addresses
    print("context %d: bucket_ms %s" % (k, " ".join("%.2f" % x for x in t)), flush=True)
    if k % 2 == 0:
        keep.append(ctx)          # stays allocated: the next context gets other memory
    else:
        del ctx
