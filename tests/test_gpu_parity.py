"""Parity of the HIP hot path with the oracle, through the C ABI (needs an MI355X)."""
import numpy as np
import pytest

from rala_amd.synth import Dataset

import parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("run_kernel", [1, 0])
@pytest.mark.parametrize("n,g,seed", [(1000, 200_000, 1), (3000, 600_000, 21), (5000, 1_000_000, 7),
                                      (2000, 1_200_000, 33), (400, 20_000, 5),
                                      (600, 60_000, 9),        # ~100x: reads beyond the 512-event cap
                                      (1500, 12_000, 3)])      # ~750x: beyond the 2048-event cap too
def test_full_path_small(hip_ctx_factory, n, g, seed, run_kernel):
    """run_kernel=1: run-space kernel (+ position-space kernel for event-dense reads);
    run_kernel=0: every read through the position-space kernel."""
    ds = Dataset(n, g, seed)
    st = parity.oracle_stages(ds)
    ctx = hip_ctx_factory()
    ctx.set_option("use_run_kernel", run_kernel)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    parity.check_initialize(ctx, st, ds)
    ctx.construct()
    parity.check_construct(ctx, st)
    parity.check_tr(ctx, st)
    tm = ctx.timings()
    print(tm)
    if run_kernel and (n, g) == (1500, 12_000):
        assert tm["pile_position_reads"] > 0 and tm["pile_overflow_reads"] >= tm["pile_position_reads"]
