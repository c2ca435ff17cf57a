"""GPU (one device): the sharded run (rala_hip_mg_*, rala_amd/csrc/sharded.hip) with its P ranks as
host threads of this process, all on device 0, exchanging through the in-process transport
(RALA_HIP_COMM_LOCAL) - the same collectives, kernels and orchestration as over RCCL, which refuses
two ranks on one device.  Every rank's replicated result against the oracle, stage by stage; and
the RCCL transport itself with a world of one."""
import numpy as np
import pytest

from rala_amd import hip
from rala_amd.synth import Dataset

import parity

pytestmark = pytest.mark.gpu


class Sharded:
    """P ranks on one device; closes everything at the end of the test"""

    def __init__(self, ds, world, device=0, token=None):
        self.ds, self.world = ds, world
        self.group = hip.LocalGroup(world) if token is None else None
        self.ranks = []
        ov = ds.overlaps
        self.cuts = hip.slice_cuts(ov.a_id, world, ov.b_id)
        for k in range(world):
            r = hip.ShardedRank(device, k, world, self.group if token is None else token)
            self.ranks.append(r)
            r.set_reads(ds.read_len)
            r.set_overlaps(ov.take(slice(self.cuts[k], self.cuts[k + 1])), self.cuts[k])

    def run(self, sens=None):
        if sens is None:
            return hip.run_ranks(self.ranks)
        n = len(sens)
        cut = [n * k // self.world for k in range(self.world + 1)]
        return hip.run_ranks(self.ranks, [sens.take(slice(cut[k], cut[k + 1])) for k in range(self.world)])

    def close(self):
        for r in self.ranks:
            r.close()
        if self.group:
            self.group.close()


@pytest.fixture
def sharded_factory():
    made = []

    def make(ds, world, **kw):
        s = Sharded(ds, world, **kw)
        made.append(s)
        return s
    yield make
    for s in made:
        s.close()


def check_rank(ctx, st, n_tr):
    """one rank's replicated result against the oracle's stages"""
    hp = ctx.piles()
    for k in ("alive", "begin", "end", "median"):
        parity.assert_same("piles2." + k, hp[k], st["piles2"][k])
    for which, key in ((0, "ov"), (1, "int")):
        h, o = ctx.overlap_list(which), st[key]
        parity.assert_same(key + ".src", h["src"], o["src"].astype(np.uint32))
        for f in ("a_begin", "a_end", "b_begin", "b_end", "length", "type"):
            parity.assert_same(key + "." + f, h[f], o[f])
    g = ctx.graph()
    parity.assert_same("nodes", g["node_read"], st["nodes"])
    for f in ("src", "dst", "len", "marked"):
        parity.assert_same("edges." + f, g[f], st["edges"][f])
    assert n_tr == st["n_tr"]


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("n,g,seed", [(1000, 200_000, 1), (2000, 1_200_000, 33), (3000, 600_000, 21), (600, 60_000, 9)])
def test_sharded_run_matches_oracle(sharded_factory, world, n, g, seed):
    ds = Dataset(n, g, seed)
    st = parity.oracle_stages(ds)
    sh = sharded_factory(ds, world)
    n_tr = sh.run()
    for r in sh.ranks:
        check_rank(r.context(), st, n_tr)
    # pits and hills as they stand after the run, hill counters included (summed over the slices)
    o = st["oracle"]
    ctx = sh.ranks[world - 1].context()
    for kind in (0, 1):
        offs, pairs, aux = ctx.intervals(kind)
        want = o.all_intervals(kind)
        parity.assert_same("intervals%d.offsets" % kind, offs, want[0])
        parity.assert_same("intervals%d.pairs" % kind, pairs, want[1])
    # the coverage of a read lives on its owner
    for r, want in list(st["data0"].items())[:30]:
        if st["piles2"]["alive"][r]:
            got = sh.ranks[r % world].pile_data(r)
            B, E = int(st["piles2"]["begin"][r]), int(st["piles2"]["end"][r])
            exp = np.array(want, copy=True)
            exp[:B] = 0
            exp[E:] = 0
            parity.assert_same("pile_data[%d]" % r, got, exp)
    tm = sh.ranks[0].timings()
    assert tm["total_ms"] > 0 and (world == 1 or tm["tuples_sent"] > 0)
    # a second run on the same objects gives the same answer
    assert sh.run() == n_tr
    check_rank(sh.ranks[0].context(), st, n_tr)


@pytest.mark.parametrize("n,g,seed,factor", [(3000, 600_000, 21, 1), (600, 60_000, 9, 1), (400, 80_000, 5, 230)])
@pytest.mark.parametrize("form", ["tuples", "records"])
def test_sharded_run_bound_tuples_instead_of_records(sharded_factory, n, g, seed, factor, form):
    """The default (every other test here): the senders scatter the bounds ONCE, by (owner, partition of the owner's reads),
    and the owners start at the second level of the partitioned bucketing (round 5; reads shorter than 2^25 - 32 bases).
    These are the older formats, by option: "records" - 8-byte records {local read, begin, end} grouped by owner only, the
    owners bucket them from the first level on (rounds 3 - 4; reads shorter than 2^21 - 32 bases); "tuples" - two tuples
    {local read, bound} per overlap side through the owners' single-pass kernel (round 2) - and what reads of 2.3 M bases
    take by themselves when the blocks are switched off."""
    from test_gpu_parity import _Scaled

    ds = Dataset(n, g, seed)
    if factor > 1:
        ds = _Scaled(ds, factor)
        assert int(ds.read_len.max()) > (1 << 21)
    st = parity.oracle_stages(ds)
    sh = sharded_factory(ds, 3)
    for r in sh.ranks:
        r.context().set_option("use_fused_emit", 0)
        if factor == 1 and form == "tuples":
            r.context().set_option("use_bound_records", 0)
    n_tr = sh.run()
    for r in sh.ranks:
        check_rank(r.context(), st, n_tr)


def test_sharded_run_long_reads_through_the_blocks(sharded_factory):
    """reads of 2.3 M bases (beyond the 21 coordinate bits of the older record format) still take the blocks: 25 bits"""
    from test_gpu_parity import _Scaled

    ds = _Scaled(Dataset(400, 80_000, 5), 230)
    st = parity.oracle_stages(ds)
    sh = sharded_factory(ds, 3)
    n_tr = sh.run()
    for r in sh.ranks:
        check_rank(r.context(), st, n_tr)


@pytest.mark.parametrize("limit", [0, 40])
def test_sharded_run_fixed_points_through_the_long_lists_kernel(sharded_factory, limit):
    """the gathered end of the second pass' fixed point and the tail's scans on every rank through the kernel for lists
    that do not fit the LDS (resident workgroups, a barrier per round): three ranks on one GPU, each with its own"""
    ds = Dataset(3000, 600_000, 21)
    st = parity.oracle_stages(ds)
    sh = sharded_factory(ds, 3)
    for r in sh.ranks:
        r.context().set_option("debug_fp_lds_limit", limit)
    n_tr = sh.run()
    for r in sh.ranks:
        check_rank(r.context(), st, n_tr)


def test_sharded_unordered_runs_and_unresolved_names(sharded_factory):
    """runs of equal queries with unresolved records inside them must not be cut (duplicate removal
    is per run, graph.cpp:343-350): shuffled runs, duplicates, names that do not resolve"""
    from rala_amd.synth import Overlaps

    ds = Dataset(1500, 300_000, 4)
    ov = ds.overlaps
    rng = np.random.default_rng(7)
    # duplicate some records right behind their original, knock out some names
    idx = np.sort(np.concatenate([np.arange(len(ov)), rng.choice(len(ov), size=len(ov) // 20, replace=False)]))
    ov2 = ov.take(idx)
    bad = rng.choice(len(ov2), size=len(ov2) // 50, replace=False)
    a = ov2.a_id.copy(); b = ov2.b_id.copy()
    a[bad[::2]] = hip.NO_READ
    b[bad[1::2]] = hip.NO_READ
    ov2 = Overlaps(a_id=a, b_id=b, a_begin=ov2.a_begin, a_end=ov2.a_end, b_begin=ov2.b_begin, b_end=ov2.b_end,
                   length=ov2.length, strand=ov2.strand)

    class DS:
        pass
    d = DS()
    d.read_len, d.overlaps, d.n_reads = ds.read_len, ov2, ds.n_reads
    st = parity.oracle_stages(d)
    for world in (2, 5):
        sh = sharded_factory(d, world)
        # no cut inside a run, unresolved records skipped when looking for the run's query
        for c in sh.cuts[1:-1]:
            j = c
            while j > 0 and a[j - 1] == hip.NO_READ:
                j -= 1
            assert c == len(a) or a[c] == hip.NO_READ or j == 0 or a[j - 1] != a[c]
        n_tr = sh.run()
        check_rank(sh.ranks[0].context(), st, n_tr)
        v = np.concatenate([r.context().valid() for r in sh.ranks])
        parity.assert_same("valid", v, st["valid"])


def test_sharded_everything_filtered_is_the_same_error_everywhere(sharded_factory):
    ds = Dataset(300, 30_000_000, 2, plants=0)    # 0.1x coverage, no planted stacks: no read has a valid region
    sh = sharded_factory(ds, 3)
    with pytest.raises(hip.RalaHipError) as e:
        sh.run()
    assert e.value.code == -4


def test_one_ranks_failure_releases_the_others(sharded_factory):
    """A failure that only ONE rank sees, between two collectives (here: at the start of the second
    overlap pass, whose first collective the others have already entered), must not leave the others
    waiting for ever: the failing rank aborts the group and every rank's run returns an error."""
    import threading

    ds = Dataset(1000, 200_000, 1)
    sh = sharded_factory(ds, 3)
    sh.ranks[1].context().set_option("debug_fail_construct", 1)
    result = {}

    def go():
        try:
            sh.run()
            result["rc"] = "ok"
        except hip.RalaHipError as e:
            result["rc"] = str(e)

    t = threading.Thread(target=go, daemon=True)
    t.start()
    t.join(120)
    assert not t.is_alive(), "the surviving ranks hang"
    assert result["rc"] != "ok"
    assert "debug_fail_construct" in result["rc"] and "aborted" in result["rc"]


def test_ranks_must_agree_on_the_sensitive_pass(sharded_factory):
    """NULL on one rank and a (possibly empty) share on another is an error on every rank, not a hang;
    an EMPTY share on some ranks is fine and gives the result of the unsharded run"""
    ds = Dataset(5000, 1_000_000, 7)
    ctx = hip.Context(0)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    ctx.construct()
    p = ctx.piles()
    sens = ds.sensitive(p["alive"], p["begin"], p["end"])
    ctx.close()
    one = hip.Context(0)
    one.set_reads(ds.read_len)
    one.set_overlaps(ds.overlaps)
    one.initialize()
    one.construct(sens)
    want = one.remove_transitive_edges()
    want_e = one.graph()
    one.close()
    sh = sharded_factory(ds, 3)
    # every sensitive overlap on rank 0, empty shares on ranks 1 and 2
    shares = [sens, sens.take(slice(0, 0)), sens.take(slice(0, 0))]
    assert hip.run_ranks(sh.ranks, shares) == want
    for r in sh.ranks:
        g = r.context().graph()
        for k in ("src", "dst", "len", "marked"):
            parity.assert_same("edges." + k, g[k], want_e[k])
    sh2 = sharded_factory(ds, 2)
    with pytest.raises(hip.RalaHipError) as e:
        hip.run_ranks(sh2.ranks, [sens, None])
    assert "sensitive" in str(e.value)


def test_threaded_runner_is_the_unsharded_result():
    """bench.py's bare `--gpus N` path (multi.ThreadedRunner: rank objects created on threads of
    their own, steps through rala_hip_mg_run_threads), here with the ranks sharing device 0"""
    from rala_amd import multi

    ds = Dataset(2000, 400_000, 11)
    st = parity.oracle_stages(ds)
    r = multi.ThreadedRunner(ds, 4, devices=[0, 0, 0, 0], transport="local")
    try:
        for _ in range(2):                       # a second step re-runs everything on the same objects
            n_tr = r.step()
            check_rank(r.ranks[0].context(), st, n_tr)
            check_rank(r.ranks[3].context(), st, n_tr)
        assert r.timings()["pile_ms"] > 0
    finally:
        r.close()
    one = multi.ThreadedRunner(ds, 1, transport="rccl")     # RCCL joined from a thread, world of one
    try:
        check_rank(one.ranks[0].context(), st, one.step())
    finally:
        one.close()


def test_sharded_run_through_rccl_world1(sharded_factory):
    """the RCCL transport (librccl opened at run time): communicator, all-to-all, all-gather(v),
    all-reduce with one rank - everything but a second GPU"""
    ds = Dataset(2000, 400_000, 11)
    st = parity.oracle_stages(ds)
    sh = sharded_factory(ds, 1, token=hip.unique_id())
    n_tr = sh.ranks[0].run()
    check_rank(sh.ranks[0].context(), st, n_tr)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("n,g,seed", [(5000, 1_000_000, 7), (6000, 1_600_000, 19)])
def test_sharded_sensitive_pass_matches_oracle(sharded_factory, world, n, g, seed):
    """Graph::preprocess with the sensitive overlaps (-s, reference graph.cpp:882-1054) in a sharded
    run: every rank holds a share of the sensitive overlaps, the target bounds go to the read
    owners in one all-to-all, medians and repeat hills come back by all-gather, bridged flags
    by all-reduce.  Every rank's result against the oracle."""
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
    want_flags = np.concatenate([o.repeat_flags(r) for r in range(n)])
    want_ov = o.overlap_list(0)
    want_p = o.piles()
    o.build_graph()
    want_tr = o.remove_transitive_edges()
    want_e = o.edges()

    sh = sharded_factory(ds, world)
    assert sh.run(sens) == want_tr
    for r in sh.ranks:
        ctx = r.context()
        offs, pairs, flags = ctx.intervals(2)
        parity.assert_same("rep.offsets", offs, want_rep[0])
        parity.assert_same("rep.pairs", pairs, want_rep[1])
        parity.assert_same("rep.flags", flags.astype(np.uint8), want_flags)
        hp = ctx.piles()
        for k in ("alive", "begin", "end", "median", "p10"):
            parity.assert_same("piles." + k, hp[k], want_p[k])
        h = ctx.overlap_list(0)
        parity.assert_same("ov.src", h["src"], want_ov["src"].astype(np.uint32))
        gr = ctx.graph()
        for k in ("src", "dst", "len", "marked"):
            parity.assert_same("edges." + k, gr[k], want_e[k])
    assert len(want_rep[1]) > 0
    # coverage of a few targets after the second add_layers, from their owners
    for t in np.unique(sens.b_id)[:24]:
        parity.assert_same("pile_data[%d]" % t, sh.ranks[int(t) % world].pile_data(int(t)), o.pile_data(int(t)))
