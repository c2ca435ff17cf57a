"""Diagnostic: is the bucketing stage's mode (2.4 or 3.3 ms at C3) a property of the process or of the
context's allocations?  Contexts one after the other in one process, each destroyed before the next
(argv[2] = 1: kept alive instead), the median bucket_ms of 15 full steps each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

ds = Dataset.config("c3")
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 4
keep_alive = len(sys.argv) > 2 and sys.argv[2] == "1"
kept = []
out = []
for k in range(n_ctx):
    ctx = hip.Context(0)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    t = []
    for _ in range(15):
        ctx.initialize()
        t.append(ctx.timings()["bucket_ms"])
        ctx.construct()
        ctx.remove_transitive_edges()
    t = sorted(t[2:])
    out.append("%.2f (%.2f - %.2f)" % (t[len(t) // 2], t[0], t[-1]))
    if keep_alive:
        kept.append(ctx)
    else:
        ctx.close()
print("pid %d, contexts %s: %s" % (os.getpid(), "kept" if keep_alive else "closed", "  ".join(out)), flush=True)
