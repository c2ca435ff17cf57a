"""Crafted inputs for SURVEY B-T1: Pile::add_layers keeps a uint16_t running coverage
(reference pile.cpp:282-288); an overlap shorter than 30 bases has its end bound (end - 15) sorted
in front of its begin bound (begin + 15), so coverage is decremented first.  Inside covered
sequence that is a dip by one; where coverage is zero it wraps to 65535 - and 65535 >= 4, so a
wrapped stretch can join two covered pieces into ONE valid region that survives
find_valid_region with the wrapped values inside.

Three kinds of target read (all survive, all 10 kb, ~40x):
  fill  two blocks of overlaps meet at a junction J (left block ends at J, right block starts
        at J - 10): after the +-15 shrink coverage is zero on [J-15, J+5); k overlaps of span 10
        on [J-10, J) put (0 - k) mod 2^16 exactly there
  dip   full-length coverage plus a dozen overlaps of span 10 .. 29 at random places
  edge  a wrapped stretch in the uncovered first 300 bases (cut away by the valid region)
"""
import numpy as np

from rala_amd.synth import Overlaps

L = 10_000


def wrap_inputs(seed=0, n_fill=12, n_dip=8, n_edge=4, depth=40):
    rng = np.random.default_rng(seed)
    n_targets = n_fill + n_dip + n_edge
    n_partners = 3 * depth
    n_reads = n_targets + n_partners
    read_len = np.full(n_reads, L, dtype=np.uint32)
    rows = []                       # (a, b, a_begin, a_end, b_begin, b_end, strand)

    def add(a, b, ab, ae, strand):
        span = ae - ab
        bb = int(rng.integers(0, L - span))
        rows.append((a, b, ab, ae, bb, bb + span, strand))

    for t in range(n_targets):
        partners = n_targets + rng.permutation(n_partners)
        p = iter(partners.tolist())
        kind = "fill" if t < n_fill else "dip" if t < n_fill + n_dip else "edge"
        mine = []
        if kind == "fill":
            J = int(rng.integers(3000, 7000))
            for _ in range(depth):
                mine.append((next(p), int(rng.integers(0, 200)), J, int(rng.integers(0, 2))))
            for _ in range(depth):
                mine.append((next(p), J - 10, int(rng.integers(L - 200, L + 1)), int(rng.integers(0, 2))))
            for _ in range(1 + t % 3):                       # k = 1, 2, 3 wrapped layers
                mine.append((next(p), J - 10, J, 0))
        elif kind == "dip":
            for _ in range(depth):
                mine.append((next(p), int(rng.integers(0, 150)), int(rng.integers(L - 150, L + 1)), int(rng.integers(0, 2))))
            for _ in range(12):
                s = int(rng.integers(10, 30))
                b0 = int(rng.integers(500, L - 500))
                mine.append((next(p), b0, b0 + s, int(rng.integers(0, 2))))
        else:
            for _ in range(depth):
                mine.append((next(p), int(rng.integers(300, 400)), int(rng.integers(L - 150, L + 1)), int(rng.integers(0, 2))))
            for _ in range(2):
                b0 = int(rng.integers(40, 200))
                mine.append((next(p), b0, b0 + int(rng.integers(10, 30)), 0))
        mine.sort(key=lambda m: m[0])                        # a run per query, sorted by target
        for b, ab, ae, strand in mine:
            add(t, b, ab, ae, strand)
    a = np.array(rows, dtype=np.int64)
    span = np.maximum(a[:, 3] - a[:, 2], a[:, 5] - a[:, 4])
    ov = Overlaps(a_id=a[:, 0], b_id=a[:, 1], a_begin=a[:, 2], a_end=a[:, 3], b_begin=a[:, 4], b_end=a[:, 5],
                  length=span, strand=a[:, 6])
    return read_len, ov, {"fill": range(0, n_fill), "dip": range(n_fill, n_fill + n_dip),
                          "edge": range(n_fill + n_dip, n_targets)}
