import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from rala_amd import hip
from rala_amd.synth import Dataset
import test_gpu_fullsize as T
wl = sys.argv[1]
ds = Dataset.config(wl)
ctx = hip.Context(0)
ctx.set_reads(ds.read_len); ctx.set_overlaps(ds.overlaps)
ctx.initialize(); ctx.construct(); print("n_tr", ctx.remove_transitive_edges())
t=time.time()
print(T._layout_tail(ctx), "%.1f s" % (time.time()-t))
