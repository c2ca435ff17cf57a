"""Comparison of an implementation (flat oracle or the HIP path) with tests/golden/*.npz.

The fixtures were produced by oracle/gen_golden.py from the REAL rala::Pile / rala::Overlap
objects (oracle/_ref).  Inputs are regenerated from the stored generator parameters and
checked against the stored SHA-256.
"""
import hashlib
import os

import numpy as np

from rala_amd.synth import Dataset, Overlaps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SETS = ("c1", "sparse", "plain", "dense")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def dataset_for(g):
    n, gl, seed, plants = (int(x) for x in g["params"])
    ds = Dataset(n, gl, seed, plants)
    m = hashlib.sha256()
    m.update(ds.read_len.tobytes())
    for a in ds.overlaps.arrays():
        m.update(a.tobytes())
    assert m.digest() == g["input_sha256"].tobytes(), "synthetic generator no longer reproduces the fixture input"
    assert len(ds.overlaps) == int(g["n_overlaps"])
    return ds


def data_digest(a):
    return int.from_bytes(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()[:8], "little")


def same(name, got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, "%s: shape %s vs golden %s" % (name, got.shape, want.shape)
    if not (got == want).all():
        bad = np.nonzero(got != want)
        first = tuple(int(x[0]) for x in bad)
        raise AssertionError("%s: %d mismatches, first at %s: got %s, golden %s" %
                             (name, len(bad[0]), first, got[first], want[first]))


def crafted_inputs():
    g = load("crafted")
    ov = Overlaps(**{k: g["in_" + k] for k in ("a_id", "b_id", "a_begin", "a_end", "b_begin", "b_end", "length",
                                               "strand")})
    return g, g["read_len"], ov
