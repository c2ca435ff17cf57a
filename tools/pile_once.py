"""Diagnostic: run Graph::initialize on one workload (for rocprofv3 --pmc passes); optional
comma-separated list of debug_pile_stop_after values, one initialize per value."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
stops = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [99, 99]
ds = Dataset.config(wl)
ctx = hip.Context(0)
ctx.set_reads(ds.read_len); ctx.set_overlaps(ds.overlaps)
for k in stops:
    ctx.set_option("debug_pile_stop_after", k)
    try:
        ctx.initialize()
    except hip.RalaHipError as e:
        if e.code != -4: raise
    print("stop", k, "pile_ms", ctx.timings()["pile_ms"])
