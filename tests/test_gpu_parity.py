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


@pytest.mark.parametrize("n,g,seed", [(3000, 600_000, 21), (2000, 1_200_000, 33), (600, 60_000, 9)])
def test_host_tail_cross_check(hip_ctx_factory, n, g, seed):
    """use_gpu_tail = 0: Graph::preprocess on the host (the path the sensitive pass uses)."""
    ds = Dataset(n, g, seed)
    st = parity.oracle_stages(ds)
    ctx = hip_ctx_factory()
    ctx.set_option("use_gpu_tail", 0)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    ctx.construct()
    parity.check_construct(ctx, st)
    parity.check_tr(ctx, st)


@pytest.mark.parametrize("n,g,seed", [(5000, 1_000_000, 7), (6000, 1_600_000, 19)])
def test_sensitive_pass_vs_oracle(hip_ctx_factory, n, g, seed):
    """Graph::preprocess(overlaps, sensitive path) (reference graph.cpp:882-1054)."""
    from oracle.oracle import Oracle

    ds = Dataset(n, g, seed)
    o = Oracle(ds.read_len, ds.overlaps, n_threads=8)
    assert o.initialize() == 0
    o.pass2()
    o.preprocess_chimeras()
    p = o.piles()
    sens = ds.sensitive(p["alive"], p["begin"], p["end"])
    o.preprocess_repeats(sens)
    want_rep = o.all_intervals(2)
    want_flags = [o.repeat_flags(r) for r in range(n)]
    want_flags = np.concatenate(want_flags) if want_flags else np.zeros(0, np.uint8)
    want_ov = o.overlap_list(0)
    want_p = o.piles()
    o.build_graph()
    want_tr = o.remove_transitive_edges()
    want_e = o.edges()

    ctx = hip_ctx_factory()
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    ctx.construct(sens)
    offs, pairs, flags = ctx.intervals(2)
    parity.assert_same("rep.offsets", offs, want_rep[0])
    parity.assert_same("rep.pairs", pairs, want_rep[1])
    parity.assert_same("rep.flags", flags.astype(np.uint8), want_flags)
    assert len(pairs) > 0, "the data set should exercise repeat hills"
    hp = ctx.piles()
    for k in ("alive", "begin", "end", "median", "p10"):
        parity.assert_same("piles." + k, hp[k], want_p[k])
    h = ctx.overlap_list(0)
    parity.assert_same("ov.src", h["src"], want_ov["src"].astype(np.uint32))
    assert ctx.remove_transitive_edges() == want_tr
    gr = ctx.graph()
    for k in ("src", "dst", "len", "marked"):
        parity.assert_same("edges." + k, gr[k], want_e[k])
    # coverage of a few targets after the second add_layers
    tg = np.unique(sens.b_id)[:16]
    for r in tg:
        parity.assert_same("pile_data[%d]" % r, ctx.pile_data(int(r)), o.pile_data(int(r)))


class _Scaled:
    """A synthetic data set with every coordinate and length multiplied by `factor`."""

    def __init__(self, ds, factor):
        from rala_amd.synth import Overlaps, FIELDS

        ov = ds.overlaps
        kw = {f: getattr(ov, f) for f in FIELDS}
        for f in ("a_begin", "a_end", "b_begin", "b_end", "length"):
            kw[f] = kw[f] * factor
        self.overlaps = Overlaps(strand=ov.strand, **kw)
        self.read_len = (ds.read_len * factor).astype(np.uint32)
        self.n_reads = ds.n_reads


@pytest.mark.parametrize("factor", [2, 7])
def test_long_reads(hip_ctx_factory, factor):
    """Reads longer than the 16384-base position bitmap (sorted-event path of the run-space
    kernel) and longer than 65536 bases (run starts beyond 16 bits)."""
    ds = _Scaled(Dataset(1500, 300_000, 11), factor)
    assert ds.read_len.max() > (16384 if factor == 2 else 65536)
    st = parity.oracle_stages(ds)
    ctx = hip_ctx_factory()
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    parity.check_initialize(ctx, st, ds)
    ctx.construct()
    parity.check_construct(ctx, st)
    parity.check_tr(ctx, st)
