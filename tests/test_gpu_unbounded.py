"""Piles whose slope-region, pit and hill lists outgrow every fixed capacity, and interval pools that turn out too small
("unbounded per-pile lists"): the reference keeps all of these in vectors (pile.cpp:66, 98, 110, 359,
448; pile.hpp:164-169), so any pile works - here the position-space kernels run such a read again with its lists in
global memory, doubled until it fits, and a pool that is too small is grown to the counted need.  Everything against
the oracle on the reference's own Pile / Overlap objects where they are built."""
import numpy as np
import pytest

from rala_amd.synth import Dataset

import parity
from sawcase import SawData

pytestmark = pytest.mark.gpu


def _ref():
    from oracle import oracle as om
    return om.have_ref()


SAW = [("pits", 600, 100, 50),          # 1 202 regions per threshold, 599 pits, 4 000 raw hills
       ("hills", 250, 120, 40),         # 502 regions at q = 1.3 only
       ("pits", 40, 300, 100),          # 82 regions, 39 pits: the LDS lists' sizes
       ("pits", 300, 3000, 300)]        # a 904 kb read: 299 pits and 300 hills that stay apart


@pytest.mark.parametrize("run_kernel,caps", [(1, 0), (0, 0), (1, (8 << 32) | 8)])
def test_saw_tooth_piles(hip_ctx_factory, run_kernel, caps):
    """more than 192 slope regions per threshold, more than 255 pits and more than 255 hills on one read, more than 64
    raw intervals before the merge - beside a generated data set, through every stage; caps: the lists in global memory
    start at 8 entries, so that the doubling runs a few times"""
    ds = SawData(SAW, base=Dataset(1000, 200_000, 1))
    st = parity.oracle_stages(ds, ref=_ref(), data_of=ds.targets)
    t0 = ds.targets[0]
    assert st["pits0"][0][t0 + 1] - st["pits0"][0][t0] > 255
    t3 = ds.targets[3]
    assert st["hills0"][0][t3 + 1] - st["hills0"][0][t3] > 255 and st["pits0"][0][t3 + 1] - st["pits0"][0][t3] > 255
    ctx = hip_ctx_factory()
    ctx.set_option("use_run_kernel", run_kernel)
    if caps:
        ctx.set_option("debug_big_caps", caps)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    tm = ctx.timings()
    assert tm["pile_unbounded_reads"] >= 3, tm
    parity.check_initialize(ctx, st, ds)
    offs, pairs, aux = ctx.intervals(0)
    for t in ds.targets:                                        # the pits' minima decide break_over_chimeric_pits
        d = st["data0"][t]
        want = [int(d[a:b + 1].min()) for a, b in pairs[offs[t]:offs[t + 1]]]
        assert aux[offs[t]:offs[t + 1]].tolist() == want, t
    ctx.construct()
    parity.check_construct(ctx, st)
    parity.check_tr(ctx, st)


@pytest.mark.parametrize("pool_x1000", [1000, 10])
def test_more_pits_and_hills_than_reads(hip_ctx_factory, pool_x1000):
    """the interval pool is sized at one slot per read (an option, now a hint): targets that share their partners have
    more pits and hills together than there are reads - the pool is grown to the counted need and the stage runs again"""
    ds = SawData([("pits", 300, 100, 50)] * 6 + [("hills", 100, 3000, 300)] * 2, share_partners=True)
    st = parity.oracle_stages(ds, ref=_ref())
    n_iv = len(st["pits0"][1]) + len(st["hills0"][1])
    assert n_iv > ds.n_reads, (n_iv, ds.n_reads)
    ctx = hip_ctx_factory()
    ctx.set_option("interval_pool_per_read_x1000", pool_x1000)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    tm = ctx.timings()
    assert tm["pool_regrown"] >= 1, tm
    parity.check_initialize(ctx, st, ds)
    ctx.construct()
    parity.check_construct(ctx, st)
    parity.check_tr(ctx, st)
    # the same context again: the pool keeps its size
    ctx.initialize()
    assert ctx.timings()["pool_regrown"] == 0
    parity.check_initialize(ctx, st, ds)


@pytest.mark.parametrize("run_kernel", [1, 0])
@pytest.mark.parametrize("n,g,seed,plants", [(6000, 14_000_000, 5, 63), (6000, 12_000_000, 6, 47), (8000, 13_000_000, 7, 63)])
def test_low_coverage_long_reads(hip_ctx_factory, n, g, seed, plants, run_kernel):
    """5 - 9x coverage, heavy-tailed lengths, one read in eighty 100 - 400 kb long ("the adversarial case
    for these caps is low coverage x long reads"): shallow piles are flagged nearly everywhere, so their regions are few
    and long (at most 48 per threshold here) - every stage against the oracle all the same"""
    ds = Dataset(n, g, seed, plants=plants)
    L = ds.read_len.astype(np.int64)
    assert (L >= 100_000).sum() >= 60 and 5.0 < L.sum() / g < 9.5
    st = parity.oracle_stages(ds, ref=_ref())
    ctx = hip_ctx_factory()
    ctx.set_option("use_run_kernel", run_kernel)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    parity.check_initialize(ctx, st, ds)
    ctx.construct()
    parity.check_construct(ctx, st)
    parity.check_tr(ctx, st)


@pytest.mark.parametrize("world", [2, 3])
def test_saw_tooth_piles_sharded(world):
    """the same piles over several ranks: the owners' lists and pools grow rank by rank, the per-read state travels with
    32-bit counts"""
    from test_gpu_sharded import Sharded, check_rank

    ds = SawData(SAW[:3] + [("pits", 300, 100, 50)] * 10, base=Dataset(1000, 200_000, 1))
    st = parity.oracle_stages(ds, ref=_ref())
    sh = Sharded(ds, world)
    try:
        from rala_amd import hip
        for r in sh.ranks:                                          # the owners' pools start at the floor: 1 024 slots
            hip.Context(_borrowed=r.L.rala_hip_mg_owner_context(r.h)).set_option("interval_pool_per_read_x1000", 10)
            r.set_reads(ds.read_len)
        n_tr = sh.run()
        assert sum(r.owner_timings()["pool_regrown"] for r in sh.ranks) >= 1
        assert sum(r.owner_timings()["pile_unbounded_reads"] for r in sh.ranks) >= 5
        for r in sh.ranks:
            check_rank(r.context(), st, n_tr)
        o = st["oracle"]
        ctx = sh.ranks[world - 1].context()
        for kind in (0, 1):
            offs, pairs, aux = ctx.intervals(kind)
            want = o.all_intervals(kind)
            parity.assert_same("intervals%d.offsets" % kind, offs, want[0])
            parity.assert_same("intervals%d.pairs" % kind, pairs, want[1])
    finally:
        sh.close()
