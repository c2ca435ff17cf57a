"""GPU: the HIP path at the full BASELINE sizes.

* against SHA-256 digests of every oracle stage (tests/golden/fullsize_<workload>.json, made by
  tests/golden/make_fullsize_digests.py on the CPU) - bit-exactness at 5 M and 50 M overlaps
  without running the oracle on the GPU box;
* through size-independent properties: the two independent pile kernels agree, coverage is
  additive (sum over a read's pile = clipped spans of its overlaps), a second transitive
  reduction finds nothing, two runs give the same result (atomics do not leak their order)."""
import hashlib
import json
import os

import numpy as np
import pytest

from rala_amd.synth import Dataset

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = 400


def dg(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def sample_reads(alive):
    rng = np.random.default_rng(12345)
    live = np.nonzero(alive)[0]
    return np.sort(rng.choice(live, size=min(SAMPLE, len(live)), replace=False))


_DS = {}


def dataset(wl):
    if wl not in _DS:
        _DS.clear()                 # one big data set at a time
        _DS[wl] = Dataset.config(wl)
    return _DS[wl]


def run(ctx_factory, ds, **options):
    ctx = ctx_factory()
    for k, v in options.items():
        ctx.set_option(k, v)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    return ctx


def stage_digests(ctx, with_data=True):
    out = {}
    p = ctx.piles()
    out["valid"] = dg(np.packbits(ctx.valid()))
    out["piles0"] = dg(*[p[k] for k in ("begin", "end", "median", "p10", "alive")])
    for kind, name in ((0, "pits0"), (1, "hills0")):
        offs, pairs, _aux = ctx.intervals(kind)
        out[name] = dg(offs.astype(np.uint64), pairs.astype(np.uint32))
    if with_data:
        reads = sample_reads(p["alive"])
        out["data_reads"] = dg(reads.astype(np.int64))
        out["data0"] = dg(*[np.asarray(ctx.pile_data(int(r)), dtype=np.uint16) for r in reads])
    # EVERY row (round 6): the per-read FNV-1a-64 of Pile::data() and the rows' sums, computed where the rows lie
    fnv, inside, outside = ctx.pile_row_digests()
    out["rows0"], out["rows0_sum"] = dg(fnv), dg(inside)
    assert not outside.any()
    ctx.construct()
    p2 = ctx.piles()
    out["piles2"] = dg(p2["begin"], p2["end"], p2["alive"])
    fnv, inside, _ = ctx.pile_row_digests()
    out["rows2"], out["rows2_sum"] = dg(fnv), dg(inside)
    for which, name in ((0, "ov"), (1, "int")):
        lst = ctx.overlap_list(which)
        out["n_%s_kept" % ("overlaps" if which == 0 else "internals")] = int(len(lst["src"]))
        out[name] = dg(*[np.asarray(lst[k]).astype(np.uint32) for k in
                         ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")])
    out["nodes"] = dg(ctx.graph()["node_read"].astype(np.uint32))
    out["n_tr"] = int(ctx.remove_transitive_edges())
    g = ctx.graph()
    out["n_edges"] = int(len(g["src"]))
    out["edges"] = dg(g["src"].astype(np.uint32), g["dst"].astype(np.uint32), g["len"].astype(np.uint32),
                      g["marked"].astype(np.uint8))
    return out


def load_digests(name):
    """digest file of the oracle; the committed ones come from the oracle running on the
    reference's own Pile / Overlap objects (make_fullsize_digests.py records the backend)"""
    path = os.path.join(HERE, "golden", "fullsize_%s.json" % name)
    if not os.path.exists(path):
        pytest.skip("no digest file %s" % name)
    want = json.load(open(path))
    assert want.get("backend") == "reference-objects", "digests must come from the reference-object oracle"
    return want


# c5x: 200 k reads at the 75x coverage of BASELINE configs[4] (15 M overlaps) - the regime in which
# the 1024- and 2048-event instantiations of the pile kernel carry a real share of the reads
@pytest.mark.parametrize("wl", ["c2", "c5x", "c3"])
def test_fullsize_matches_oracle_digests(hip_ctx_factory, wl):
    want = load_digests(wl)
    ds = dataset(wl)
    assert (ds.n_reads, len(ds.overlaps)) == (want["n_reads"], want["n_overlaps"])
    ctx = run(hip_ctx_factory, ds)
    tm = ctx.timings()
    got = stage_digests(ctx)
    for k, v in got.items():
        assert v == want[k], "%s: stage %s differs from the oracle" % (wl, k)
    if wl == "c5x":
        assert tm["pile_overflow_reads"] > 500, tm          # beyond the 512-event instantiation


@pytest.mark.parametrize("wl", ["c2", "c3"])
def test_fullsize_properties(hip_ctx_factory, wl):
    ds = dataset(wl)
    ov = ds.overlaps
    ctx = run(hip_ctx_factory, ds)
    p = ctx.piles()
    pits, hills = ctx.intervals(0), ctx.intervals(1)
    reads = sample_reads(p["alive"])[:100]
    data = {int(r): np.asarray(ctx.pile_data(int(r)), dtype=np.int64) for r in reads}

    # 1. additivity: inside the valid region the pile is the sum of the (shrunk) overlap spans
    sel = np.isin(ov.a_id, reads) | np.isin(ov.b_id, reads)
    idx = np.nonzero(sel)[0]
    for r in reads:
        r = int(r)
        B, E = int(p["begin"][r]), int(p["end"][r])
        total = 0
        for side_id, sb, se in ((ov.a_id, ov.a_begin, ov.a_end), (ov.b_id, ov.b_begin, ov.b_end)):
            m = idx[side_id[idx] == r]
            lo = np.clip(sb[m].astype(np.int64) + 15, B, E)
            hi = np.clip(se[m].astype(np.int64) - 15, B, E)
            total += int(np.maximum(hi - lo, 0).sum())
        assert int(data[r][B:E].sum()) == total, r
        assert data[r][:B].sum() == 0 and data[r][E:].sum() == 0
        assert (data[r][B:E] >= 4).all()                 # Pile::find_valid_region

    # 2. the position-space kernel (an independent implementation) agrees
    other = run(hip_ctx_factory, ds, use_run_kernel=0)
    q = other.piles()
    for k in ("begin", "end", "median", "p10", "alive"):
        assert (p[k] == q[k]).all(), k
    for mine, theirs in ((pits, other.intervals(0)), (hills, other.intervals(1))):
        assert (mine[0] == theirs[0]).all() and (mine[1] == theirs[1]).all()
    for r in reads[:20]:
        assert (data[int(r)] == np.asarray(other.pile_data(int(r)), dtype=np.int64)).all()
    del other

    # 3. a second transitive reduction on the reduced graph marks nothing
    ctx.construct()
    n_tr = ctx.remove_transitive_edges()
    g = ctx.graph()
    assert n_tr > 0 and int(g["marked"].sum()) == 2 * n_tr
    keep = g["marked"] == 0
    marks, pairs = ctx.tr_mark(len(g["node_read"]), g["src"][keep], g["dst"][keep], g["len"][keep])
    assert pairs == 0 and not marks.any()

    # 4. determinism: a second context gives the same graph
    again = run(hip_ctx_factory, ds)
    again.construct()
    assert again.remove_transitive_edges() == n_tr
    h = again.graph()
    for k in ("node_read", "src", "dst", "len", "marked"):
        assert (g[k] == h[k]).all(), k


@pytest.mark.parametrize("wl,world", [("c2", 4), ("c3", 8)])
def test_fullsize_sharded_run(wl, world):
    """The sharded run (rala_hip_mg_*: slices, owners, collectives through the in-process transport,
    the ranks as host threads on this one GPU) at full size: every stage digest of every rank's
    replicated result equals the oracle's."""
    from test_gpu_sharded import Sharded

    want = load_digests(wl)
    ds = dataset(wl)
    sh = Sharded(ds, world)
    try:
        n_tr = sh.run()
        assert n_tr == want["n_tr"]
        valid = np.concatenate([r.context().valid() for r in sh.ranks])
        assert dg(np.packbits(valid)) == want["valid"]
        for r in (sh.ranks[0], sh.ranks[world - 1]):
            ctx = r.context()
            got = {}
            p2 = ctx.piles()
            got["piles2"] = dg(p2["begin"], p2["end"], p2["alive"])
            for which, name in ((0, "ov"), (1, "int")):
                lst = ctx.overlap_list(which)
                got["n_%s_kept" % ("overlaps" if which == 0 else "internals")] = int(len(lst["src"]))
                got[name] = dg(*[np.asarray(lst[k]).astype(np.uint32) for k in
                                 ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")])
            g = ctx.graph()
            got["nodes"] = dg(g["node_read"].astype(np.uint32))
            got["n_edges"] = int(len(g["src"]))
            got["edges"] = dg(g["src"].astype(np.uint32), g["dst"].astype(np.uint32), g["len"].astype(np.uint32),
                              g["marked"].astype(np.uint8))
            for k, v in got.items():
                assert v == want[k], "%s sharded over %d ranks: stage %s differs from the oracle" % (wl, world, k)
        # every row from its owner, under the final regions
        fnv = np.zeros(ds.n_reads, dtype=np.uint64)
        tot = np.zeros(ds.n_reads, dtype=np.uint64)
        for k, r in enumerate(sh.ranks):
            fnv[k::world], tot[k::world], _ = r.pile_row_digests()
        assert dg(fnv) == want["rows2"] and dg(tot) == want["rows2_sum"]
        # coverage vectors come from their owners: against a single-context run
        reads = sample_reads(sh.ranks[0].context().piles()["alive"])[:40]
        single = run(lambda: __import__("rala_amd.hip", fromlist=["Context"]).Context(0), ds)
        single.construct()
        for r in reads:
            assert (sh.ranks[int(r) % world].pile_data(int(r)) == single.pile_data(int(r))).all(), int(r)
        single.close()
        print(wl, world, sh.ranks[0].timings())
    finally:
        sh.close()


@pytest.mark.skipif(os.environ.get("RALA_SKIP_C5") == "1", reason="RALA_SKIP_C5=1")
def test_c5_properties():
    """the largest BASELINE configuration on one GPU, one context at a time (a C5 context holds
    80 GB of piles + 32 GB of bound slots): additivity, agreement of the two pile kernels,
    idempotent transitive reduction, run-to-run determinism"""
    from rala_amd import hip

    ds = dataset("c5")
    ov = ds.overlaps

    def one(**options):
        ctx = hip.Context(0)
        try:
            for k, v in options.items():
                ctx.set_option(k, v)
            ctx.set_reads(ds.read_len)
            ctx.set_overlaps(ov)
            ctx.initialize()
            p = ctx.piles()
            pits, hills = ctx.intervals(0), ctx.intervals(1)
            reads = sample_reads(p["alive"])[:60]
            data = {int(r): np.asarray(ctx.pile_data(int(r)), dtype=np.int64) for r in reads}
            rows = ctx.pile_row_digests()               # every one of the 4 M rows, hashed and summed where it lies
            ctx.construct()
            n_tr = ctx.remove_transitive_edges()
            g = ctx.graph()
            res = dict(p=p, pits=pits, hills=hills, reads=reads, data=data, n_tr=n_tr, g=g,
                       ov=ctx.overlap_list(0), tm=ctx.timings(), rows=rows)
            if not options:
                keep = g["marked"] == 0
                marks, pairs = ctx.tr_mark(len(g["node_read"]), g["src"][keep], g["dst"][keep], g["len"][keep])
                res["second_tr"] = (int(pairs), bool(marks.any()))
            return res
        finally:
            ctx.close()

    a = one()
    p, reads, data = a["p"], a["reads"], a["data"]
    assert a["tm"]["pile_overflow_reads"] > 1000           # the 1024-event instantiation is exercised
    # additivity on the sampled reads
    sel = np.nonzero(np.isin(ov.a_id, reads) | np.isin(ov.b_id, reads))[0]
    for r in reads:
        r = int(r)
        B, E = int(p["begin"][r]), int(p["end"][r])
        total = 0
        for side_id, sb, se in ((ov.a_id, ov.a_begin, ov.a_end), (ov.b_id, ov.b_begin, ov.b_end)):
            m = sel[side_id[sel] == r]
            lo = np.clip(sb[m].astype(np.int64) + 15, B, E)
            hi = np.clip(se[m].astype(np.int64) - 15, B, E)
            total += int(np.maximum(hi - lo, 0).sum())
        assert int(data[r][B:E].sum()) == total, r
        assert data[r][:B].sum() == 0 and data[r][E:].sum() == 0
    assert a["n_tr"] > 0 and a["second_tr"] == (0, False)
    assert int(a["g"]["marked"].sum()) == 2 * a["n_tr"]
    # additivity over ALL 4 M reads (no host holds C5's reference objects): the sum of every row over its valid region, taken
    # on the device, is the sum of the clipped spans of the read's overlaps (Pile::add_layers' +-15, reference
    # graph.cpp:311-326: the bounds of every record that resolved, duplicates included); nothing is stored outside the region
    fnv, inside, outside = a["rows"]
    assert not outside.any() and not fnv[p["alive"] == 0].any() and fnv[p["alive"] != 0].all()
    B, E = p["begin"].astype(np.int64), p["end"].astype(np.int64)
    want = np.zeros(ds.n_reads, dtype=np.float64)
    step = 1 << 25
    for lo_i in range(0, len(ov), step):
        sl = slice(lo_i, min(len(ov), lo_i + step))
        for side_id, sb, se in ((ov.a_id, ov.a_begin, ov.a_end), (ov.b_id, ov.b_begin, ov.b_end)):
            ids = side_id[sl].astype(np.int64)
            lo = np.clip(sb[sl].astype(np.int64) + 15, B[ids], E[ids])
            hi = np.clip(se[sl].astype(np.int64) - 15, B[ids], E[ids])
            want += np.bincount(ids, weights=np.maximum(hi - lo, 0).astype(np.float64), minlength=ds.n_reads)
    alive = p["alive"] != 0
    assert (inside[alive].astype(np.float64) == want[alive]).all(), int(np.nonzero(alive & (inside.astype(np.float64) != want))[0][0])

    def same(x, y, what):
        for k in ("begin", "end", "median", "p10", "alive"):
            assert (x["p"][k] == y["p"][k]).all(), (what, k)
        for name in ("pits", "hills"):
            assert (x[name][0] == y[name][0]).all() and (x[name][1] == y[name][1]).all(), (what, name)
        for r in reads[:20]:
            assert (x["data"][int(r)] == y["data"][int(r)]).all(), (what, int(r))
        for k in range(3):          # every row: the two pile kernels (and every bucketing path) leave the same 80 GB
            assert (x["rows"][k] == y["rows"][k]).all(), (what, "rows", k)
        assert x["n_tr"] == y["n_tr"], what
        for k in ("node_read", "src", "dst", "len", "marked"):
            assert (x["g"][k] == y["g"][k]).all(), (what, k)
        for k in ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type"):
            assert (np.asarray(x["ov"][k]) == np.asarray(y["ov"][k])).all(), (what, k)

    same(a, one(use_run_kernel=0), "position-space kernel")
    same(a, one(use_fixed_buckets=0), "exact CSR bucketing")
    same(a, one(use_partitioned_buckets=0), "single-pass bucketing into fixed slots")
    same(a, one(use_round_batches=0), "every containment round looked at by the host")
    same(a, one(), "second run")


def _sens_result(ctx, n_tr=None):
    """everything the sensitive pass leaves behind, as digests; n_tr: what the (sharded) run returned, or
    None to run the transitive reduction here"""
    out = {}
    offs, pairs, flags = ctx.intervals(2)
    out["rep"] = dg(offs.astype(np.uint64), pairs.astype(np.uint32), flags.astype(np.uint8))
    out["n_repeat_hills"] = int(len(pairs))
    p3 = ctx.piles()
    out["piles3"] = dg(*[p3[k] for k in ("begin", "end", "median", "p10", "alive")])
    ov = ctx.overlap_list(0)
    out["n_overlaps_kept_sens"] = int(len(ov["src"]))
    out["ov_sens"] = dg(*[np.asarray(ov[k]).astype(np.uint32) for k in
                          ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")])
    out["n_tr"] = int(ctx.remove_transitive_edges()) if n_tr is None else int(n_tr)
    g = ctx.graph()
    out["nodes"] = dg(g["node_read"].astype(np.uint32))
    out["n_edges"] = int(len(g["src"]))
    out["edges"] = dg(g["src"].astype(np.uint32), g["dst"].astype(np.uint32), g["len"].astype(np.uint32),
                      g["marked"].astype(np.uint8))
    return out, p3


@pytest.mark.skipif(os.environ.get("RALA_SKIP_C5") == "1", reason="RALA_SKIP_C5=1")
def test_c5_sensitive_pass_and_sharded_run():
    """BASELINE configs[4] as far as one GPU goes: `-s` at 4 M reads / 300 M overlaps (the sensitive set derived
    from the HIP piles, as the two-pass workflow of the reference derives it from the first pass), and the
    sharded runner over 8 ranks at that size (in-process transport, all ranks on this GPU), primary and `-s`.
    No oracle can hold C5 here: the checks are the second add_layers' additivity on sampled targets, the
    run-space against the position-space kernels, run-to-run determinism, and every rank of the sharded run
    against the single-context result."""
    from rala_amd import hip
    from test_gpu_sharded import Sharded

    ds = dataset("c5")
    ov = ds.overlaps
    world = 8

    def single(sens=None, keep=None, **options):
        ctx = hip.Context(0)
        try:
            for k, v in options.items():
                ctx.set_option(k, v)
            ctx.set_reads(ds.read_len)
            ctx.set_overlaps(ov)
            ctx.initialize()
            if sens is None:
                ctx.construct()
                p2 = ctx.piles()
                return p2, None, None
            before = {int(t): np.asarray(ctx.pile_data(int(t)), dtype=np.int64) for t in keep} if keep is not None else None
            ctx.construct(sens)
            res, p3 = _sens_result(ctx)
            after = {int(t): np.asarray(ctx.pile_data(int(t)), dtype=np.int64) for t in keep} if keep is not None else None
            return res, p3, (before, after)
        finally:
            ctx.close()

    # the chimera stage alone gives the piles the generator derives the sensitive set from
    p2, _, _ = single()
    sens = ds.sensitive(p2["alive"], p2["begin"], p2["end"])
    assert len(sens) > 10_000_000, len(sens)
    targets = np.unique(sens.b_id)
    rng = np.random.default_rng(7)
    sample = np.sort(rng.choice(targets, size=48, replace=False))

    want, p3, (before, after) = single(sens, keep=sample)
    assert want["n_repeat_hills"] > 0 and want["n_tr"] > 0
    # Pile::add_layers of the sensitive bounds (graph.cpp:929-947: target side only, no +-15, shifted by the
    # target's valid-region begin) on top of the first pass' coverage: what a sampled target gained inside
    # its valid region is the clipped spans of the sensitive overlaps that hit it
    sel = np.nonzero(np.isin(sens.b_id, sample))[0]
    for t in sample:
        t = int(t)
        B, E = int(p2["begin"][t]), int(p2["end"][t])
        m = sel[sens.b_id[sel] == t]
        lo = np.clip(sens.b_begin[m].astype(np.int64) + B, B, E)
        hi = np.clip(sens.b_end[m].astype(np.int64) + B, B, E)
        gained = int(np.maximum(hi - lo, 0).sum())
        assert int(after[t][B:E].sum()) - int(before[t][B:E].sum()) == gained, t
        assert after[t][:B].sum() == 0 and after[t][E:].sum() == 0

    # the position-space kernels (pile_kernels.hip, pile_repeats_kernel.hip) agree with the run-space ones
    other, _, (_, after_pos) = single(sens, keep=sample[:12], use_run_kernel=0)
    assert other == want, {k: (other[k], want[k]) for k in want if other[k] != want[k]}
    for t in sample[:12]:
        assert (after_pos[int(t)] == after[int(t)]).all(), int(t)
    # determinism
    again, _, _ = single(sens)
    assert again == want

    # the sharded runner at this size: 8 ranks, primary pass and -s, every rank's replicated result
    sh = Sharded(ds, world)
    try:
        n = len(sens)
        cut = [n * k // world for k in range(world + 1)]
        shares = [sens.take(slice(cut[k], cut[k + 1])) for k in range(world)]
        n_tr = hip.run_ranks(sh.ranks, shares)
        assert n_tr == want["n_tr"]
        for r in (sh.ranks[0], sh.ranks[3], sh.ranks[world - 1]):
            got, _ = _sens_result(r.context(), n_tr)
            assert got == want, {k: (got[k], want[k]) for k in want if got[k] != want[k]}
        for t in sample[:16]:
            assert (np.asarray(sh.ranks[int(t) % world].pile_data(int(t)), dtype=np.int64) == after[int(t)]).all(), int(t)
        print("c5 sharded over", world, "ranks:", sh.ranks[0].timings())
        # BASELINE configs[4]'s last clause at its stated size: "full layout" - the clean-up stages behind the transitive
        # reduction (Graph::simplify, reference graph.cpp:642-697: tips, bubbles, five rounds of the force-directed layout
        # with remove_long_edges, graph.cpp:1056-1279) and create_unitigs, on the graph the 8 ranks built - twice
        first = _layout_tail(sh.ranks[world - 1].context())
        print("c5 layout tail:", first)
        # (at 75x, behind the sensitive pass, the synthetic genome comes out as ONE chain of 281 k reads - no tip, bubble or
        # long edge is left to find, the layout still runs over a component of that size; the stages that find them are
        # compared with the oracle at the sizes it can hold, tests/test_gpu_cli.py, and at C3 they all fire)
        assert first["largest_component"] >= 100_000 and first["contigs"] >= 1 and first["unitigs"] >= 1, first
        again = _layout_tail(sh.ranks[0].context())
        assert again == first, (again, first)
        print("c5 layout tail:", first)
    finally:
        sh.close()


def _layout_tail(ctx):
    """The host remainder of Graph behind the transitive reduction (rala_amd/host/assembly_graph.cpp through its C API) on a
    context's graph, the layout steps on the GPU (rala_hip_layout).  The reads' sequences are stand-ins of the right
    length (bases from a generator seeded with the read id): what the stages look at is lengths, the digests cover the
    data all the same."""
    import hashlib
    import time

    import layout

    t0 = time.time()
    g = ctx.graph()
    p = ctx.piles()
    G = layout.product()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for r in g["node_read"][::2]:
        r = int(r)
        n = int(p["end"][r]) - int(p["begin"][r])
        data = acgt[np.random.default_rng(r).integers(0, 4, size=n, dtype=np.uint8)].tobytes()
        G.add_node_pair(r, b"r%d" % r, data)
    for s_, d_, l_ in zip(g["src"].tolist(), g["dst"].tolist(), g["len"].tolist()):
        G.add_edge(s_, d_, l_)
    for i in np.nonzero(g["marked"])[0]:
        if i % 2 == 0:
            G.mark_edge(int(i))
    G.note_transitive()
    G.remove_marked(False)
    t1 = time.time()
    # components of what is left, for the record: the layout kernel is O(points^2) per component
    nodes, edges = G.dump()
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csg
    live = edges["alive"] != 0
    nn = len(nodes["alive"])
    adj = sp.coo_matrix((np.ones(int(live.sum()), np.uint8), (edges["begin"][live] // 2, edges["end"][live] // 2)), shape=(nn // 2, nn // 2))
    _, labels = csg.connected_components(adj, directed=False)
    sizes = np.bincount(labels)
    count = {"tips": 0, "bubbles": 0, "long_edges": 0, "largest_component": int(sizes.max()), "nodes": int(nn),
             "edges_after_tr": int(live.sum())}

    def engine(x, y, adj_off, adj_, iterations, k, t, dt):
        ctx.layout(x, y, adj_off, adj_, iterations, k, t, dt)
        return 0

    def loop():
        while True:
            t, b = G.run("tips"), G.run("bubbles")
            count["tips"] += t
            count["bubbles"] += b
            if t + b == 0:
                break
    loop()
    G.run("shrink", 42)
    for seed in range(5):
        G.postprocess(seed, engine)
        count["long_edges"] += G.run("long_edges")
        count["tips"] += G.run("tips")
    loop()
    t2 = time.time()
    count["unitigs"] = G.run("unitigs")
    nodes, edges = G.dump()
    alive = nodes["alive"] != 0
    count["contigs"] = int((alive & (nodes["n_seq"] > 1)).sum()) // 2
    h = hashlib.sha256()
    for k in ("alive", "length", "n_seq", "data_hash", "ids_hash"):
        h.update(np.ascontiguousarray(nodes[k]).tobytes())
    for k in ("alive", "begin", "end", "length"):
        h.update(np.ascontiguousarray(edges[k]).tobytes())
    count["digest"] = h.hexdigest()[:16]
    print("layout tail: build %.1f s, simplify %.1f s, unitigs %.1f s" % (t1 - t0, t2 - t1, time.time() - t2))
    return count


@pytest.mark.parametrize("wl", ["c2", "c5x", "c3"])
def test_fullsize_sensitive_pass_matches_oracle_digests(hip_ctx_factory, wl):
    """Graph::preprocess with the sensitive overlap set (-s) at full size: repeat hills and their
    bridged flags, piles (second add_layers + medians), the filtered overlap list, graph and
    transitive reduction against the oracle's digests (tests/golden/fullsize_<wl>_sens.json)"""
    want = load_digests(wl + "_sens")
    ds = dataset(wl)
    ctx = run(hip_ctx_factory, ds)
    ctx.construct()                                          # the chimera stage alone gives the piles ...
    p2 = ctx.piles()
    assert dg(p2["begin"], p2["end"], p2["alive"]) == want["piles2"]
    sens = ds.sensitive(p2["alive"], p2["begin"], p2["end"])     # ... the generator derives the set from
    assert len(sens) == want["n_sensitive"]
    ctx.initialize()
    ctx.construct(sens)
    got = {}
    offs, pairs, flags = ctx.intervals(2)
    got["rep"] = dg(offs.astype(np.uint64), pairs.astype(np.uint32), flags.astype(np.uint8))
    got["n_repeat_hills"] = int(len(pairs))
    p3 = ctx.piles()
    got["piles3"] = dg(*[p3[k] for k in ("begin", "end", "median", "p10", "alive")])
    targets = np.unique(sens.b_id)[:SAMPLE]
    got["data3"] = dg(*[np.asarray(ctx.pile_data(int(r)), dtype=np.uint16) for r in targets])
    fnv, inside, _ = ctx.pile_row_digests()           # every row, the targets' with their second add_layers
    got["rows3"], got["rows3_sum"] = dg(fnv), dg(inside)
    ov = ctx.overlap_list(0)
    got["n_overlaps_kept_sens"] = int(len(ov["src"]))
    got["ov_sens"] = dg(*[np.asarray(ov[k]).astype(np.uint32) for k in
                          ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")])
    got["nodes"] = dg(ctx.graph()["node_read"].astype(np.uint32))
    got["n_tr"] = int(ctx.remove_transitive_edges())
    g = ctx.graph()
    got["n_edges"] = int(len(g["src"]))
    got["edges"] = dg(g["src"].astype(np.uint32), g["dst"].astype(np.uint32), g["len"].astype(np.uint32),
                      g["marked"].astype(np.uint8))
    for k, v in got.items():
        assert v == want[k], "%s: stage %s differs from the oracle" % (wl, k)
