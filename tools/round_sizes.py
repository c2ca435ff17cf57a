import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rala_amd import hip
from rala_amd.synth import Dataset
ds = Dataset.config("c3")
ctx = hip.Context(0)
ctx.set_option("use_round_batches", 0)
ctx.set_reads(ds.read_len); ctx.set_overlaps(ds.overlaps)
ctx.initialize(); ctx.construct()
print(ctx.timings())
