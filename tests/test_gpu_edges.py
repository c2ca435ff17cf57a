"""Edge inputs through the C ABI against the oracle (needs an MI355X): nothing to do, nothing left, one of everything,
records that resolve to nothing - single context and sharded."""
import numpy as np
import pytest

from rala_amd import hip
from rala_amd.synth import Dataset, Overlaps, FIELDS

import parity

pytestmark = pytest.mark.gpu

NO_READ = 0xFFFFFFFF


class _Set:
    """a data set given as arrays"""
    def __init__(self, read_len, **cols):
        self.read_len = np.ascontiguousarray(read_len, dtype=np.uint32)
        self.n_reads = len(self.read_len)
        n = len(cols.get("a_id", []))
        full = {f: np.asarray(cols.get(f, np.zeros(n)), dtype=np.uint32) for f in FIELDS}
        self.overlaps = Overlaps(**full, strand=np.asarray(cols.get("strand", np.zeros(n)), dtype=np.uint8))


def _both(hip_ctx_factory, ds):
    """the oracle's verdict and, if it goes on, every stage; the same from the device"""
    st = parity.oracle_stages(ds)
    ctx = hip_ctx_factory()
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    if st["init_rc"] != 0:
        with pytest.raises(hip.RalaHipError) as e:
            ctx.initialize()
        assert e.value.code == -4            # RALA_HIP_EFILTERED: "filtered all sequences"
        return st, ctx
    ctx.initialize()
    parity.check_initialize(ctx, st, ds)
    ctx.construct()
    parity.check_construct(ctx, st)
    parity.check_tr(ctx, st)
    return st, ctx


def test_no_overlaps_at_all(hip_ctx_factory):
    st, _ = _both(hip_ctx_factory, _Set([5000, 8000, 12000]))
    assert st["init_rc"] != 0


def test_one_read_without_overlaps(hip_ctx_factory):
    st, _ = _both(hip_ctx_factory, _Set([9000]))
    assert st["init_rc"] != 0


def test_records_that_resolve_to_nothing(hip_ctx_factory):
    """every record names an unknown read on one side or both"""
    n = 500
    rng = np.random.default_rng(1)
    a = np.sort(rng.integers(0, 40, n)).astype(np.uint32)
    b = rng.integers(0, 40, n).astype(np.uint32)
    a[::2] = NO_READ
    b[1::2] = NO_READ
    ds = _Set([10000] * 40, a_id=a, b_id=b, a_begin=np.full(n, 100), a_end=np.full(n, 9000), b_begin=np.full(n, 50),
              b_end=np.full(n, 8950), length=np.full(n, 8950))
    st, _ = _both(hip_ctx_factory, ds)
    assert st["init_rc"] != 0


def test_reads_too_short_to_keep(hip_ctx_factory):
    """full-length overlaps between reads of 1 200 bases: no valid region reaches the 1 260 bases of pile.cpp:307"""
    n_reads, n = 30, 600
    rng = np.random.default_rng(2)
    a = np.sort(rng.integers(0, n_reads - 1, n)).astype(np.uint32)
    b = (a + 1 + rng.integers(0, n_reads - 1, n) % (n_reads - 1 - a)).astype(np.uint32)
    ds = _Set([1200] * n_reads, a_id=a, b_id=b, a_begin=np.zeros(n), a_end=np.full(n, 1200), b_begin=np.zeros(n),
              b_end=np.full(n, 1200), length=np.full(n, 1200))
    st, _ = _both(hip_ctx_factory, ds)
    assert st["init_rc"] != 0


def test_only_self_overlaps(hip_ctx_factory):
    n_reads = 20
    a = np.repeat(np.arange(n_reads, dtype=np.uint32), 8)
    n = len(a)
    ds = _Set([10000] * n_reads, a_id=a, b_id=a.copy(), a_begin=np.full(n, 10), a_end=np.full(n, 9990), b_begin=np.full(n, 10),
              b_end=np.full(n, 9990), length=np.full(n, 9980))
    _both(hip_ctx_factory, ds)


def test_two_reads_one_overlap_many_times(hip_ctx_factory):
    """the same dovetail between two reads, written ten times: a pile of coverage 10 on both, duplicates removed for the
    graph - whatever that leaves (oracle's word)"""
    n = 10
    ds = _Set([10000, 10000], a_id=np.zeros(n), b_id=np.ones(n), a_begin=np.full(n, 4000), a_end=np.full(n, 10000),
              b_begin=np.zeros(n), b_end=np.full(n, 6000), length=np.full(n, 6000))
    _both(hip_ctx_factory, ds)


@pytest.mark.parametrize("n,g,seed,plants", [(64, 12_000, 3, 0), (129, 30_000, 4, 15), (300, 3_000_000, 5, 15), (2, 3_000, 6, 0)])
def test_small_and_sparse_sets(hip_ctx_factory, n, g, seed, plants):
    """a handful of reads; a genome so large that hardly two reads meet (most reads lose everything); two reads"""
    _both(hip_ctx_factory, Dataset(n, g, seed, plants))


@pytest.mark.parametrize("n,g,seed", [(40, 60_000, 8), (24, 30_000, 9), (16, 20_000, 8), (12, 2_000, 8)])
def test_sharded_with_ranks_that_own_or_see_next_to_nothing(n, g, seed):
    """eight ranks and a few dozen reads: a rank owns two to five reads, the slices of some ranks are empty; a set of which
    nothing is left"""
    from test_gpu_sharded import Sharded, check_rank

    ds = Dataset(n, g, seed, 0)
    st = parity.oracle_stages(ds)
    sh = Sharded(ds, 8)
    try:
        if st["init_rc"] != 0:
            with pytest.raises(hip.RalaHipError):
                sh.run()
            return
        n_tr = sh.run()
        for r in sh.ranks:
            check_rank(r.context(), st, n_tr)
    finally:
        sh.close()
