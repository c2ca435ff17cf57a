"""Differential fuzzing of the HIP path against the oracle on random synthetic data sets
(python tests/fuzz_parity.py [n_cases] [first_seed]); prints one line per case, exits 1 on a
mismatch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from rala_amd import hip
from rala_amd.synth import Dataset
import parity

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for case in range(n_cases):
    seed = first + case
    rng = np.random.default_rng(seed)
    n = int(rng.integers(300, 5000))
    cov = float(rng.choice([8, 20, 35, 60, 110]))
    g = int(max(30_000, n * 10_000 / cov))
    plants = int(rng.integers(0, 16))
    try:
        ds = Dataset(n, g, seed, plants)
        st = parity.oracle_stages(ds, n_threads=os.cpu_count() or 8)
        ctx = hip.Context(0)
        ctx.set_option("use_run_kernel", int(rng.random() < 0.85))
        ctx.set_option("use_fixed_buckets", int(rng.random() < 0.8))
        ctx.set_reads(ds.read_len)
        ctx.set_overlaps(ds.overlaps)
        try:
            ctx.initialize()
            rc = 0
        except hip.RalaHipError as e:
            rc = e.code
        if st["init_rc"] != 0:
            assert rc == -4, (rc, st["init_rc"])
            print("case %d n=%d g=%d cov=%g plants=%d: all filtered (both)" % (seed, n, g, cov, plants), flush=True)
            continue
        assert rc == 0, rc
        parity.check_initialize(ctx, st, ds)
        ctx.construct()
        parity.check_construct(ctx, st)
        parity.check_tr(ctx, st)
        tm = ctx.timings()
        print("case %d n=%d g=%d cov=%g plants=%d: ok (%d overlaps, %d kept, overflow %d, position %d)" % (
            seed, n, g, cov, plants, len(ds.overlaps), len(st["ov"]["src"]), tm["pile_overflow_reads"],
            tm["pile_position_reads"]), flush=True)
        ctx.close()
    except AssertionError as e:
        bad += 1
        print("case %d n=%d g=%d cov=%g plants=%d: MISMATCH %s" % (seed, n, g, cov, plants, e), flush=True)
print("%d mismatches in %d cases" % (bad, n_cases))
sys.exit(1 if bad else 0)
