"""Stage-by-stage comparison of the HIP path (through the C ABI) with the oracle."""
import numpy as np

from oracle.oracle import Oracle


def oracle_stages(ds, n_threads=8, ref=False, with_tr=True, n_data=64, data_of=()):
    """Run the oracle on a synthetic data set, snapshotting every stage."""
    o = Oracle(ds.read_len, ds.overlaps, n_threads=n_threads, ref=ref)
    st = {}
    rc = o.initialize()
    st["init_rc"] = rc
    st["valid"] = o.valid()
    st["piles0"] = o.piles()
    st["pits0"] = o.all_intervals(0)
    st["hills0"] = o.all_intervals(1)
    st["oracle"] = o
    st["rows0"] = o.pile_row_digests()       # (fnv, sum) of EVERY pile's data_ (round 6)
    # coverage vectors of a sample of live reads and of every read with pits / hills
    alive = np.nonzero(st["piles0"]["alive"])[0]
    rng = np.random.default_rng(0)
    pick = set(rng.choice(alive, size=min(n_data, len(alive)), replace=False).tolist()) if len(alive) else set()
    for key in ("pits0", "hills0"):
        offs = st[key][0]
        pick.update(int(x) for x in np.nonzero(offs[1:] > offs[:-1])[0][:32])
    pick.update(int(r) for r in data_of)
    st["data0"] = {r: o.pile_data(r) for r in sorted(pick)}
    if rc != 0:
        return st
    o.pass2()
    o.preprocess_chimeras()
    st["piles2"] = o.piles()
    st["ov"] = o.overlap_list(0)
    st["int"] = o.overlap_list(1)
    o.build_graph()
    st["nodes"] = o.nodes()
    if with_tr:
        st["n_tr"] = o.remove_transitive_edges()
    st["edges"] = o.edges()
    return st


def assert_same(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (name, a.shape, b.shape)
    if not (a == b).all():
        bad = np.nonzero(a != b)
        first = tuple(int(x[0]) for x in bad)
        raise AssertionError("%s: %d mismatches, first at %s: hip=%s oracle=%s" %
                             (name, len(bad[0]), first, a[first], b[first]))


def check_initialize(ctx, st, ds):
    """HIP state after rala_hip_initialize against the oracle snapshot."""
    assert_same("valid", ctx.valid(), st["valid"])
    hp = ctx.piles()
    op = st["piles0"]
    for k in ("alive", "begin", "end", "median", "p10"):
        assert_same("piles0." + k, hp[k], op[k])
    for kind, key in ((0, "pits0"), (1, "hills0")):
        offs, pairs, aux = ctx.intervals(kind)
        assert_same(key + ".offsets", offs, st[key][0])
        assert_same(key + ".pairs", pairs, st[key][1])
    for r, want in st["data0"].items():
        assert_same("pile_data[%d]" % r, ctx.pile_data(r), want)
    # every row, hashed and summed where it lies (rala_hip_get_pile_row_digests) against the same vectors from the oracle's objects
    if "rows0" in st:
        fnv, inside, outside = ctx.pile_row_digests()
        assert_same("rows0.fnv", fnv, st["rows0"][0])
        assert_same("rows0.sum", inside, st["rows0"][1])
        assert not outside.any(), "values stored outside a valid region right after initialize"


def check_construct(ctx, st):
    hp = ctx.piles()
    op = st["piles2"]
    for k in ("alive", "begin", "end", "median"):
        assert_same("piles2." + k, hp[k], op[k])
    for which, key in ((0, "ov"), (1, "int")):
        h = ctx.overlap_list(which)
        o = st[key]
        assert_same(key + ".src", h["src"], o["src"].astype(np.uint32))
        for f in ("a_begin", "a_end", "b_begin", "b_end", "length", "type"):
            assert_same(key + "." + f, h[f], o[f])
    g = ctx.graph()
    assert_same("nodes", g["node_read"], st["nodes"])
    for f in ("src", "dst", "len"):
        assert_same("edges." + f, g[f], st["edges"][f])


def check_tr(ctx, st):
    n = ctx.remove_transitive_edges()
    g = ctx.graph()
    assert n == st["n_tr"], (n, st["n_tr"])
    assert_same("edges.marked", g["marked"], st["edges"]["marked"])


def run_hip(ctx, ds, construct=True):
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    if construct:
        ctx.construct()
