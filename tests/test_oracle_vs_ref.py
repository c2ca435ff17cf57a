"""CPU, only where oracle/_ref exists (this container): the flat restatement against the
REAL rala::Pile / rala::Overlap objects on random inputs, method by method."""
import numpy as np
import pytest

from oracle import oracle as om
from oracle.oracle import Oracle
from rala_amd.synth import Dataset

pytestmark = pytest.mark.skipif(not om.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def random_pile(rng, n):
    """piecewise-constant coverage with noise, dips and bumps of assorted widths"""
    base = int(rng.integers(8, 80))
    d = np.full(n, base, dtype=np.int64)
    for _ in range(int(rng.integers(0, 12))):
        a = int(rng.integers(0, n - 10))
        w = int(rng.choice([20, 100, 300, 700, 900, 1500, 3000]))
        d[a:a + w] += int(rng.integers(-base, 3 * base))
    if rng.random() < 0.5:
        d += rng.integers(-2, 3, size=n)
    d = np.clip(d, 0, 60000)
    if rng.random() < 0.7:
        d[: int(rng.integers(0, 400))] = 0
        d[n - int(rng.integers(1, 400)):] = 0
    return d.astype(np.uint16)


def pair(n_reads, read_len):
    return (Oracle(read_len, None, ref=False), Oracle(read_len, None, ref=True))


@pytest.mark.parametrize("seed", range(6))
def test_pile_methods_fuzz(seed):
    rng = np.random.default_rng(1000 + seed)
    n_piles = 40
    lens = rng.integers(1300, 9000, size=n_piles).astype(np.uint32)
    flat, ref = pair(n_piles, lens)
    for r in range(n_piles):
        d = random_pile(rng, int(lens[r]))
        for o in (flat, ref):
            o.set_pile_state(r, d, 0, int(lens[r]))
        ok_f, ok_r = flat.find_valid_region(r), ref.find_valid_region(r)
        assert ok_f == ok_r
        if not ok_f:
            continue
        for o in (flat, ref):
            o.find_median(r)
        for q in (1.3, 1.82, 1.42, 1.1, 2.5):
            a, b = flat.find_slopes(r, q), ref.find_slopes(r, q)
            assert a.shape == b.shape and (a == b).all(), (seed, r, q)
        for o in (flat, ref):
            o.find_chimeric_hills(r)
            o.find_chimeric_pits(r)
        med = int(rng.integers(1, 120))
        for o in (flat, ref):
            o.find_repetitive_hills(r, med)
        for kind in (0, 1, 2):
            a, b = flat.intervals(r, kind), ref.intervals(r, kind)
            assert a.shape == b.shape and (a == b).all(), (seed, r, kind)
        assert (flat.pile_data(r) == ref.pile_data(r)).all()
        # breaking over pits with a random median, possibly twice
        for _ in range(2):
            m = int(rng.integers(1, 200))
            bf, br = flat.break_over_chimeric_pits(r, m), ref.break_over_chimeric_pits(r, m)
            assert bf == br
            if not bf:
                break
            a, b = flat.intervals(r, 0), ref.intervals(r, 0)
            assert a.shape == b.shape and (a == b).all()
    pf, pr = flat.piles(), ref.piles()
    for k in pf:
        assert (pf[k] == pr[k]).all(), k


@pytest.mark.parametrize("seed", range(4))
def test_add_layers_fuzz(seed):
    rng = np.random.default_rng(2000 + seed)
    lens = np.full(8, 5000, dtype=np.uint32)
    flat, ref = pair(8, lens)
    for r in range(8):
        for _ in range(3):                       # several chunks on top of each other
            k = int(rng.integers(0, 300))
            b = rng.integers(0, 4970, size=k)
            e = np.minimum(b + rng.integers(1, 3000, size=k), 5000)      # includes spans < 30 after the shrink
            bounds = np.concatenate([(b + 15) << 1, ((np.maximum(e, 15) - 15) << 1) | 1]).astype(np.uint32)
            bounds = bounds[(bounds >> 1) <= 5000]
            rng.shuffle(bounds)
            flat.add_layers(r, bounds)
            ref.add_layers(r, bounds)
        assert (flat.pile_data(r) == ref.pile_data(r)).all()


@pytest.mark.parametrize("seed", range(4))
def test_trim_type_fuzz(seed):
    rng = np.random.default_rng(3000 + seed)
    n = 30
    lens = rng.integers(2000, 9000, size=n).astype(np.uint32)
    flat, ref = pair(n, lens)
    for r in range(n):
        L = int(lens[r])
        d = np.full(L, 20, dtype=np.uint16)
        b = int(rng.integers(0, L // 4))
        e = int(rng.integers(3 * L // 4, L + 1))
        for o in (flat, ref):
            o.set_pile_state(r, d, b, e)
    for _ in range(4000):
        a, b = (int(x) for x in rng.choice(n, size=2, replace=False))
        la, lb = int(lens[a]), int(lens[b])
        mode = rng.random()
        if mode < 0.5:      # near-equal spans, all sorts of positions
            span = int(rng.integers(50, min(la, lb)))
            ab = int(rng.integers(0, la - span + 1)); bb = int(rng.integers(0, lb - span + 1))
            ae = ab + span; be = bb + span + int(rng.integers(-3, 4))
            be = max(bb + 1, min(lb, be))
        else:
            ab = int(rng.integers(0, la - 1)); ae = int(rng.integers(ab + 1, la + 1))
            bb = int(rng.integers(0, lb - 1)); be = int(rng.integers(bb + 1, lb + 1))
        length = max(ae - ab, be - bb) if rng.random() < 0.8 else int(rng.integers(1, 9000))
        strand = int(rng.integers(0, 2))
        c = [ab, ae, bb, be, length]
        rf, rr = flat.overlap_trim_type(a, b, strand, c), ref.overlap_trim_type(a, b, strand, c)
        assert rf[0] == rr[0]
        if rf[0]:
            assert (rf[1] == rr[1]).all() and rf[2] == rr[2], (a, b, strand, c, rf, rr)


@pytest.mark.parametrize("n,g,seed,sens", [(1200, 240_000, 77, True), (2500, 2_000_000, 78, False),
                                            (700, 35_000, 79, True)])
def test_whole_path_on_reference_objects(n, g, seed, sens):
    """same orchestration, flat restatement vs reference objects, stage by stage"""
    ds = Dataset(n, g, seed)
    res = []
    for ref in (False, True):
        o = Oracle(ds.read_len, ds.overlaps, n_threads=4, ref=ref)
        st = [o.initialize(), o.valid(), o.piles(), o.all_intervals(0), o.all_intervals(1)]
        o.pass2()
        st += [o.overlap_list(0), o.overlap_list(1), [o.hill_counts(r) for r in range(n)]]
        o.preprocess_chimeras()
        p = o.piles()
        st += [p, o.overlap_list(0)]
        if sens:
            s = ds.sensitive(p["alive"], p["begin"], p["end"])
            o.preprocess_repeats(s)
            st += [o.all_intervals(2), [o.repeat_flags(r) for r in range(n)], o.overlap_list(0), o.piles()]
        o.build_graph()
        st += [o.nodes(), o.remove_transitive_edges(), o.edges()]
        res.append(st)

    def eq(x, y, path):
        if isinstance(x, dict):
            for k in x:
                eq(x[k], y[k], path + "/" + k)
        elif isinstance(x, (list, tuple)):
            assert len(x) == len(y), path
            for i, (u, v) in enumerate(zip(x, y)):
                eq(u, v, "%s/%d" % (path, i))
        elif isinstance(x, np.ndarray):
            assert x.shape == y.shape and (x == y).all(), path
        else:
            assert x == y, path
    eq(res[0], res[1], "")
