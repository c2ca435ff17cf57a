"""Crafted piles whose slope-region and interval lists outgrow any fixed capacity ("unbounded per-pile
lists"): the reference keeps them in vectors (pile.cpp:66, 98, 110, 359, 448; pile.hpp:164-169), so any pile works.

A saw-tooth target: a floor of `low` overlaps over the whole read plus, every `period` bases, `high - low` short overlaps
that cover `width` bases (after the +-15 shrink of graph.cpp:317-324).  Every stretch at the floor lies within 847 bases
of a tooth on both sides, so all of its positions are flagged "down" AND "up" (floor * q < high): one down region and one
up region per tooth and threshold, resolved into (down [s, s], up [s, e]) - a pit per tooth at q = 1.82 when high > int(low *
1.82), and for q = 1.3 every (up, later down) pair within 840 bases a hill candidate.

kinds of target:
  pits    low 4, high 8: regions, pits and raw hills all grow with the number of teeth
  hills   low 4, high 6: regions and raw hills at q = 1.3 only (6 <= int(4 * 1.82) = 7: no pits)
The teeth's partners are short reads of their own that nobody else overlaps (they do not survive find_valid_region);
the targets can be appended to a generated data set (`base`) so that the later stages have something to do.
"""
import numpy as np

from rala_amd.synth import Overlaps


def saw_inputs(targets, base=None, seed=0, share_partners=False):
    """targets: list of (kind, n_teeth, period, width); returns read_len, overlaps, ids of the targets.
    share_partners: every target draws its partners from the same reads (more pits and hills than reads)"""
    rng = np.random.default_rng(seed)
    if base is not None:
        read_len = [int(x) for x in base.read_len]
        cols = [np.asarray(c, dtype=np.int64) for c in base.overlaps.arrays()]
        rows = list(zip(*[c.tolist() for c in cols[:6]], cols[7].tolist()))
    else:
        read_len, rows = [], []
    ids = []
    partner_len = 3000
    shared = []
    for kind, n_teeth, period, width in targets:
        low, high = (4, 8) if kind == "pits" else (4, 6)
        L = 2000 + n_teeth * period + 2000
        t = len(read_len)
        read_len.append(L)
        ids.append(t)
        mine, theirs = [], []           # records with the target as query / as target
        used = [0]
        def partner():
            if share_partners:
                if used[0] == len(shared):
                    shared.append(None)
                used[0] += 1
                return -used[0]                 # (ids behind all targets, fixed below)
            read_len.append(partner_len)
            return len(read_len) - 1
        for _ in range(low):            # the floor: whole read (bounds at 15 and L - 15)
            p = partner()
            mine.append((t, p, 0, L, 0, min(L, partner_len), 0))
        for k in range(n_teeth):
            s = 2000 + k * period
            for j in range(high - low):
                p = partner()
                b0 = int(rng.integers(0, partner_len - width - 30))
                rec_q = (s - 15, s + width + 15, b0, b0 + width + 30)
                if (k + j) % 2 == 0:
                    mine.append((t, p, rec_q[0], rec_q[1], rec_q[2], rec_q[3], int(rng.integers(0, 2))))
                else:
                    theirs.append((p, t, rec_q[2], rec_q[3], rec_q[0], rec_q[1], int(rng.integers(0, 2))))
        mine.sort(key=lambda m: m[1])
        rows.extend(mine)               # a run per query
        theirs.sort(key=lambda m: m[0])
        rows.extend(theirs)
    a = np.array(rows, dtype=np.int64).reshape(-1, 7)
    if share_partners:
        first = len(read_len)
        read_len.extend([partner_len] * len(shared))
        for c in (0, 1):
            neg = a[:, c] < 0
            a[neg, c] = first - 1 - a[neg, c]
        # a run per query, sorted by target inside
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
    span = np.maximum(a[:, 3] - a[:, 2], a[:, 5] - a[:, 4])
    ov = Overlaps(a_id=a[:, 0], b_id=a[:, 1], a_begin=a[:, 2], a_end=a[:, 3], b_begin=a[:, 4], b_end=a[:, 5],
                  length=span, strand=a[:, 6])
    return np.array(read_len, dtype=np.uint32), ov, ids


class SawData:
    """duck-typed like rala_amd.synth.Dataset for tests/parity.py"""

    def __init__(self, targets, base=None, seed=0, share_partners=False):
        self.read_len, self.overlaps, self.targets = saw_inputs(targets, base, seed, share_partners)
        self.n_reads = len(self.read_len)
