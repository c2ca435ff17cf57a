"""Measurement (round 6): contexts created and closed after each other with the rows in mapped chunks of 2 MB (the range released,
reserved and mapped again and again, two data sets alternating): do the rows read by a KERNEL (rala_hip_get_pile_row_digests) and by
hipMemcpy (rala_hip_get_pile_data) always agree with the first time the data set was seen?

    python tools/chunk_stress.py [iterations] [chunk MB]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sets = [Dataset(600, 60_000, 9), Dataset(9000, 1_800_000, 4), Dataset(3000, 600_000, 21)]
ref = {}
bad_kernel = bad_copy = 0
for it in range(iters):
    k = it % len(sets) if it % 7 else (it // 7) % len(sets)
    ds = sets[k]
    ctx = hip.Context(0)
    ctx.set_option("pile_chunk_mb", mb)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    p = ctx.piles()
    fnv, inside, outside = ctx.pile_row_digests()
    rows = {}
    for r in range(0, ds.n_reads, 1 if ds.n_reads <= 1000 else max(1, ds.n_reads // 150)):
        row = np.asarray(ctx.pile_data(r), dtype=np.int64)
        rows[r] = int(row[int(p["begin"][r]):int(p["end"][r])].sum())
    if k not in ref:
        ref[k] = (fnv.copy(), inside.copy())
    if not (fnv == ref[k][0]).all():
        bad_kernel += 1
        print("iteration %d, set %d: %d rows differ by the kernel's hash" % (it, k, int((fnv != ref[k][0]).sum())), flush=True)
    wrong = [r for r, s in rows.items() if p["alive"][r] and s != int(ref[k][1][r])]
    if wrong:
        bad_copy += 1
        print("iteration %d, set %d: %d of %d rows differ as copied by hipMemcpy (first: read %d)" % (it, k, len(wrong), len(rows), wrong[0]), flush=True)
    ctx.close()
print("%d iterations, chunks of %d MB: %d with rows wrong by the kernel's hash, %d with rows wrong as copied" % (iters, mb, bad_kernel, bad_copy))
