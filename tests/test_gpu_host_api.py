"""The class interface of the host mirror - stand-alone rala::Pile / rala::Overlap objects, each
pile computed by the HIP kernel through its own one-read context - driven method by method the
way Graph::initialize / preprocess do (reference src/graph.cpp:311-326, 387-407, 700-722), and
compared with the oracle (the reference's own objects where oracle/_ref is built)."""
import ctypes
import os

import numpy as np
import pytest

from rala_amd.synth import Dataset

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p
U64 = ctypes.c_uint64
U32 = ctypes.c_uint32


def _lib():
    from rala_amd import build

    build.build_host()
    L = ctypes.CDLL(os.path.join(ROOT, "rala_amd", "host", "librala_api.so"))
    L.hp_create.restype = P
    L.hp_create.argtypes = [P, U64]
    L.hp_destroy.argtypes = [P]
    L.hp_add_layers.argtypes = [P, U64, P, U64]
    for f in ("hp_find_valid_region", "hp_break_over_chimeric_hills"):
        getattr(L, f).argtypes = [P, U64]
        getattr(L, f).restype = ctypes.c_int
    for f in ("hp_find_median", "hp_find_chimeric_hills", "hp_find_chimeric_pits", "hp_reset"):
        getattr(L, f).argtypes = [P, U64]
        getattr(L, f).restype = None
    L.hp_break_over_chimeric_pits.argtypes = [P, U64, ctypes.c_uint16]
    L.hp_break_over_chimeric_pits.restype = ctypes.c_int
    L.hp_get.argtypes = [P, U64, P]
    L.hp_data.argtypes = [P, U64, P]
    L.hp_data.restype = U64
    L.hp_overlap_trim_type.argtypes = [P, U32, U32, U32, P, P]
    L.hp_overlap_trim_type.restype = ctypes.c_int
    L.hp_overlap_from_mhap.argtypes = [U64, U64] + [U32] * 8 + [P]
    L.hp_find_repetitive_hills.argtypes = [P, U64, ctypes.c_uint16]
    L.hp_find_repetitive_hills.restype = None
    L.hp_has_repetitive_hills.argtypes = [P, U64]
    L.hp_has_repetitive_hills.restype = ctypes.c_int
    L.hp_to_json.argtypes = [P, U64, ctypes.c_char_p, U64]
    L.hp_to_json.restype = U64
    return L


class HostPiles:
    def __init__(self, L, read_len):
        self.L = L
        self.len = np.ascontiguousarray(read_len, dtype=np.uint32)
        self.h = L.hp_create(self.len.ctypes.data, len(self.len))

    def close(self):
        if self.h:
            self.L.hp_destroy(self.h)
            self.h = None

    def add_layers(self, r, bounds):
        b = np.ascontiguousarray(bounds, dtype=np.uint32)
        self.L.hp_add_layers(self.h, r, b.ctypes.data, len(b))

    def get(self, r):
        out = np.zeros(7, dtype=np.uint32)
        self.L.hp_get(self.h, r, out.ctypes.data)
        return dict(zip(("alive", "begin", "end", "median", "p10", "has_pit", "has_hill"), out.tolist()))

    def data(self, r):
        out = np.zeros(int(self.len[r]), dtype=np.uint16)
        n = self.L.hp_data(self.h, r, out.ctypes.data)
        return out[:n]

    def to_json(self, r):
        n = int(self.L.hp_to_json(self.h, r, None, 0))
        buf = ctypes.create_string_buffer(n + 1)
        self.L.hp_to_json(self.h, r, buf, n)
        return buf.raw[:n].decode()

    def trim_type(self, a, b, strand, coords):
        c = np.ascontiguousarray(coords, dtype=np.uint32).copy()
        t = ctypes.c_int(-1)
        ok = self.L.hp_overlap_trim_type(self.h, a, b, strand, c.ctypes.data, ctypes.byref(t))
        assert ok >= 0, "createOverlap / transmute changed ids or strand"
        return bool(ok), c, t.value


def read_bounds(ds):
    """store_overlap_bounds (graph.cpp:311-326): per read, the two ends of every overlap, shrunk
    by 15 -> list of arrays [k, 2] (begin bound, end bound)"""
    ov = ds.overlaps
    reads = np.concatenate([ov.a_id, ov.b_id]).astype(np.int64)
    lo = np.concatenate([(ov.a_begin + 15) << 1, (ov.b_begin + 15) << 1]).astype(np.uint32)
    hi = np.concatenate([((ov.a_end - 15) << 1) | 1, ((ov.b_end - 15) << 1) | 1]).astype(np.uint32)
    order = np.argsort(reads, kind="stable")
    reads, pairs = reads[order], np.stack([lo[order], hi[order]], axis=1)
    offs = np.searchsorted(reads, np.arange(ds.n_reads + 1))
    return [pairs[offs[r]:offs[r + 1]] for r in range(ds.n_reads)]


@pytest.mark.parametrize("n,g,seed", [(250, 50_000, 3), (200, 12_000, 8)])
def test_pile_objects_follow_the_reference_method_by_method(n, g, seed):
    from oracle import oracle as om
    from oracle.oracle import Oracle

    ds = Dataset(n, g, seed)
    bounds = read_bounds(ds)
    o = Oracle(ds.read_len, None, ref=om.have_ref())
    L = _lib()
    hp = HostPiles(L, ds.read_len)
    rng = np.random.default_rng(seed)
    alive = np.zeros(n, dtype=bool)
    n_pit = n_hill = 0
    try:
        for r in range(n):
            b = bounds[r][rng.permutation(len(bounds[r]))]
            cut = len(b) // 3                    # two chunks of whole overlaps, like two parser rounds
            for part in (b[:cut], b[cut:]):
                part = part.reshape(-1).copy()
                rng.shuffle(part)
                o.add_layers(r, part.copy())
                hp.add_layers(r, part.copy())
            ok_o = o.find_valid_region(r)
            ok_h = bool(L.hp_find_valid_region(hp.h, r))
            assert ok_o == ok_h, r
            if not ok_o:
                L.hp_reset(hp.h, r)
                assert hp.get(r)["alive"] == 0
                continue
            alive[r] = True
            o.find_median(r); L.hp_find_median(hp.h, r)
            o.find_chimeric_hills(r); L.hp_find_chimeric_hills(hp.h, r)
            o.find_chimeric_pits(r); L.hp_find_chimeric_pits(hp.h, r)
        want = o.piles()
        for r in np.nonzero(alive)[0]:
            r = int(r)
            got = hp.get(r)
            for k in ("begin", "end", "median", "p10"):
                assert got[k] == int(want[k][r]), (r, k)
            assert got["has_pit"] == (len(o.intervals(r, 0)) > 0), r
            assert got["has_hill"] == (len(o.intervals(r, 1)) > 0), r
            n_pit += got["has_pit"]; n_hill += got["has_hill"]
            assert (hp.data(r) == o.pile_data(r)).all(), r
        assert alive.sum() > n // 2 and n_pit > 0

        # Graph::preprocess (graph.cpp:700-722): break over pits with the data set median, then hills
        med = int(np.median(want["median"][alive]))
        for r in np.nonzero(alive)[0]:
            r = int(r)
            for m in (med, 4 * med):             # a second, more aggressive round
                bo, bh = o.break_over_chimeric_pits(r, m), bool(L.hp_break_over_chimeric_pits(hp.h, r, m))
                assert bo == bh, (r, m)
                if not bo:
                    alive[r] = False
                    L.hp_reset(hp.h, r)
                    break
                got = hp.get(r)
                assert got["has_pit"] == (len(o.intervals(r, 0)) > 0), (r, m)
            if not alive[r]:
                continue
            bo, bh = o.break_over_chimeric_hills(r), bool(L.hp_break_over_chimeric_hills(hp.h, r))
            assert bo == bh, r
            if not bo:
                alive[r] = False
                L.hp_reset(hp.h, r)
        want = o.piles()
        for r in np.nonzero(alive)[0]:
            r = int(r)
            got = hp.get(r)
            assert (got["begin"], got["end"]) == (int(want["begin"][r]), int(want["end"][r])), r
            assert got["has_hill"] == 0
            assert (hp.data(r) == o.pile_data(r)).all(), r

        # Pile::find_repetitive_hills (pile.cpp:500-566) with the data set median, on the stand-alone
        # piles; then Pile::to_json (pile.cpp:632-663) - coverage, region, repeat hills, median, p10
        n_rep = 0
        for r in np.nonzero(alive)[0][:400]:
            r = int(r)
            for m in (med, max(1, med // 2), 1):        # (a data set median of 1 makes every bump a hill)
                o.find_repetitive_hills(r, m)
                L.hp_find_repetitive_hills(hp.h, r, m)
                want_h = o.intervals(r, 2)
                assert bool(L.hp_has_repetitive_hills(hp.h, r)) == (len(want_h) > 0), (r, m)
            js = hp.to_json(r)
            assert js == o.to_json(r), r
            n_rep += len(want_h) > 0
            import json as _json
            d = _json.loads("{" + js + "}")[str(r)]           # what misc/plotter.py reads: y, b, e, h, m, p10
            assert len(d["y"]) == int(ds.read_len[r]) and d["b"] == hp.get(r)["begin"] and len(d["h"]) == 2 * len(want_h)
        assert n_rep > 0 or n < 250

        # Overlap::transmute / trim / type against the piles as they are now
        ov = ds.overlaps
        pick = rng.choice(len(ov), size=min(len(ov), 3000), replace=False)
        kinds = set()
        for i in pick:
            a, b, s = int(ov.a_id[i]), int(ov.b_id[i]), int(ov.strand[i])
            c = [int(ov.a_begin[i]), int(ov.a_end[i]), int(ov.b_begin[i]), int(ov.b_end[i]), int(ov.length[i])]
            if not (alive[a] and alive[b]):
                assert hp.trim_type(a, b, s, c)[0] is False
                continue
            ro, rh = o.overlap_trim_type(a, b, s, c), hp.trim_type(a, b, s, c)
            assert ro[0] == rh[0], (a, b, c)
            if ro[0]:
                assert (ro[1] == rh[1]).all() and ro[2] == rh[2], (a, b, s, c, ro, rh)
                kinds.add(ro[2])
        assert len(kinds) >= 3
    finally:
        hp.close()


def test_saw_tooth_pile_objects():
    """stand-alone Pile objects whose lists outgrow every fixed capacity (tests/sawcase.py): more than 192 slope regions,
    more than 255 pits / hills, and - with a data set median of 1 - repeat hills from 7 000 candidate pairs (pile.cpp:500-566:
    every (up, later down) pair); chimeric lists through break_over_*, repeat hills through to_json"""
    from oracle import oracle as om
    from oracle.oracle import Oracle
    from sawcase import SawData

    ds = SawData([("pits", 300, 100, 50), ("hills", 130, 120, 40), ("pits", 120, 3000, 300)])
    bounds = read_bounds(ds)
    o = Oracle(ds.read_len, None, ref=om.have_ref())
    L = _lib()
    hp = HostPiles(L, ds.read_len)
    try:
        for r in ds.targets:
            part = bounds[r].reshape(-1).copy()
            o.add_layers(r, part.copy())
            hp.add_layers(r, part.copy())
            assert o.find_valid_region(r) and L.hp_find_valid_region(hp.h, r)
            o.find_median(r); L.hp_find_median(hp.h, r)
            o.find_chimeric_hills(r); L.hp_find_chimeric_hills(hp.h, r)
            o.find_chimeric_pits(r); L.hp_find_chimeric_pits(hp.h, r)
        want = o.piles()
        n_pits = [len(o.intervals(r, 0)) for r in ds.targets]
        n_hills = [len(o.intervals(r, 1)) for r in ds.targets]
        assert n_pits[0] > 255 and n_hills[2] > 100, (n_pits, n_hills)
        for r in ds.targets:
            got = hp.get(r)
            for k in ("begin", "end", "median", "p10"):
                assert got[k] == int(want[k][r]), (r, k)
            assert got["has_pit"] == (len(o.intervals(r, 0)) > 0) and got["has_hill"] == (len(o.intervals(r, 1)) > 0), r
            assert (hp.data(r) == o.pile_data(r)).all(), r
        n_rep = []
        for r in ds.targets:
            for m in (4, 1):
                o.find_repetitive_hills(r, m)
                L.hp_find_repetitive_hills(hp.h, r, m)
                assert hp.to_json(r) == o.to_json(r), (r, m)
            n_rep.append(len(o.intervals(r, 2)))
        assert max(n_rep) > 0, n_rep
        # the pits decide where the reads break (the floor is 4: a median of 8 makes every pit real)
        for r in ds.targets:
            bo, bh = o.break_over_chimeric_pits(r, 8), bool(L.hp_break_over_chimeric_pits(hp.h, r, 8))
            assert bo == bh, r
            if bo:
                bo, bh = o.break_over_chimeric_hills(r), bool(L.hp_break_over_chimeric_hills(hp.h, r))
                assert bo == bh, r
        want = o.piles()
        for r in ds.targets:
            got = hp.get(r)
            assert got["alive"] == int(want["alive"][r]), r
            if got["alive"]:
                assert (got["begin"], got["end"]) == (int(want["begin"][r]), int(want["end"][r])), r
    finally:
        hp.close()


def test_mhap_record_fields():
    """createOverlap(MHAP) (reference src/overlap.cpp:12-20): ids are 1-based in the file"""
    L = _lib()
    out = np.zeros(4, dtype=np.uint32)
    L.hp_overlap_from_mhap(7, 12, 0, 100, 4100, 9000, 1, 50, 4300, 8000, out.ctypes.data)
    assert out.tolist() == [6, 11, 4250, 1]
    L.hp_overlap_from_mhap(1, 2, 1, 0, 500, 600, 1, 10, 400, 700, out.ctypes.data)
    assert out.tolist() == [0, 1, 500, 0]
