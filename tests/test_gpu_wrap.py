"""GPU: uint16 wrap-around of the coverage (SURVEY B-T1, reference pile.cpp:282-288) on the HIP
path: overlaps shorter than 30 bases INSIDE surviving reads - dips by one in covered sequence and
stretches of (0 - k) mod 2^16 that join two covered pieces into one valid region.  Full coverage
vectors of every target read, annotations, overlap lists and the graph against the oracle running
on the reference's own Pile / Overlap objects (oracle/_ref) where that library travelled, on the
flat restatement otherwise; through both pile kernels."""
import numpy as np
import pytest

from oracle import oracle as ora
import parity
import wrapcase

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("run_kernel", [1, 0])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_wrap_inside_surviving_reads(hip_ctx_factory, run_kernel, seed):
    read_len, ov, kinds = wrapcase.wrap_inputs(seed=seed)
    o = ora.Oracle(read_len, ov, n_threads=4, ref=ora.have_ref())
    assert o.initialize() == 0
    want_p = o.piles()
    n_targets = len(kinds["fill"]) + len(kinds["dip"]) + len(kinds["edge"])
    want_d = [o.pile_data(t) for t in range(n_targets)]
    want_iv = [o.all_intervals(0), o.all_intervals(1)]
    want_valid = o.valid()

    ctx = hip_ctx_factory()
    ctx.set_option("use_run_kernel", run_kernel)
    ctx.set_reads(read_len)
    ctx.set_overlaps(ov)
    ctx.initialize()
    parity.assert_same("valid", ctx.valid(), want_valid)
    got_p = ctx.piles()
    for k in ("alive", "begin", "end", "median", "p10"):
        parity.assert_same("piles." + k, got_p[k], want_p[k])
    n_wrapped = 0
    for t in range(n_targets):
        assert want_p["alive"][t]
        d = ctx.pile_data(t)
        parity.assert_same("pile_data[%d]" % t, d, want_d[t])
        n_wrapped += int((d >= 65000).any())
    assert n_wrapped == len(kinds["fill"])          # the wrapped stretches are really there
    for kind in (0, 1):
        offs, pairs, _aux = ctx.intervals(kind)
        parity.assert_same("intervals%d.offsets" % kind, offs, want_iv[kind][0])
        parity.assert_same("intervals%d.pairs" % kind, pairs, want_iv[kind][1])

    # ... and on through the second pass, the preprocess tail, the graph and its reduction
    o.pass2()
    o.preprocess_chimeras()
    o.build_graph()
    st = {"piles2": o.piles(), "ov": o.overlap_list(0), "int": o.overlap_list(1), "nodes": o.nodes()}
    st["n_tr"] = o.remove_transitive_edges()
    st["edges"] = o.edges()
    ctx.construct()
    parity.check_construct(ctx, st)
    parity.check_tr(ctx, st)
