"""ctypes wrapper over the synthetic read/overlap generator (synth.cpp).

Bench/test input plumbing — not part of the product path.
"""
import ctypes
import os

import numpy as np

from .. import build as _build

_lib = None

# BASELINE.json configs -> (n_reads, genome_len, seed); SURVEY.md §8(d)
CONFIGS = {
    "c1": (1000, 200_000, 1),
    "c2": (100_000, 20_000_000, 2),
    "c3": (1_000_000, 200_000_000, 3),
    "c5": (4_000_000, 533_000_000, 5),
    # C5's coverage regime (4 M x 10 kb / 533 Mb = 75x) at a size a CPU run can hold: parity-test case
    "c5x": (200_000, 26_650_000, 5),
}


def lib():
    global _lib
    if _lib is None:
        path = _build.build_synth()
        L = ctypes.CDLL(path)
        L.synth_create.restype = ctypes.c_void_p
        L.synth_create.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
        L.synth_destroy.argtypes = [ctypes.c_void_p]
        for f in ("synth_n_reads", "synth_n_overlaps", "synth_n_sensitive"):
            getattr(L, f).restype = ctypes.c_uint64
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.synth_read_len.restype = ctypes.POINTER(ctypes.c_uint32)
        L.synth_read_len.argtypes = [ctypes.c_void_p]
        L.synth_read_kind.restype = ctypes.POINTER(ctypes.c_uint8)
        L.synth_read_kind.argtypes = [ctypes.c_void_p]
        L.synth_field.restype = ctypes.POINTER(ctypes.c_uint32)
        L.synth_field.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.synth_strand.restype = ctypes.POINTER(ctypes.c_uint8)
        L.synth_strand.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.synth_make_sensitive.restype = ctypes.c_uint64
        L.synth_make_sensitive.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.synth_write_fasta.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.synth_write_paf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p]
        _lib = L
    return _lib


FIELDS = ("a_id", "b_id", "a_begin", "a_end", "b_begin", "b_end", "length")


class Overlaps:
    """Structure-of-arrays overlap records (numpy, uint32 / uint8)."""

    def __init__(self, **kw):
        for f in FIELDS:
            setattr(self, f, np.ascontiguousarray(kw[f], dtype=np.uint32))
        self.strand = np.ascontiguousarray(kw["strand"], dtype=np.uint8)

    def __len__(self):
        return int(self.a_id.shape[0])

    def take(self, idx):
        return Overlaps(**{f: getattr(self, f)[idx] for f in FIELDS}, strand=self.strand[idx])

    def arrays(self):
        return [getattr(self, f) for f in FIELDS] + [self.strand]


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Dataset:
    """One synthetic data set: read lengths + primary overlaps (+ sensitive set on request)."""

    def __init__(self, n_reads, genome_len, seed, plants=15):
        """plants: bit 0 chimeras, 1 adapters, 2 repeats, 3 stacks (15: SURVEY.md Appendix E); + 16: heavy-tailed read
        lengths (3 kb + exponential, 2 % of the reads 30 - 90 kb longer) instead of N(10 000, 1 500); + 32: one read in eighty
        100 - 400 kb long"""
        L = lib()
        self._h = L.synth_create(n_reads, genome_len, seed, plants)
        self.n_reads = int(L.synth_n_reads(self._h))
        self.read_len = _np(L.synth_read_len(self._h), self.n_reads, np.uint32)
        self.read_kind = _np(L.synth_read_kind(self._h), self.n_reads, np.uint8)
        n = int(L.synth_n_overlaps(self._h))
        self.overlaps = Overlaps(**{f: _np(L.synth_field(self._h, 0, i), n, np.uint32) for i, f in enumerate(FIELDS)},
                                 strand=_np(L.synth_strand(self._h, 0), n, np.uint8))

    @classmethod
    def config(cls, name, plants=15):
        n, g, s = CONFIGS[name]
        return cls(n, g, s, plants)

    def sensitive(self, alive, begin, end):
        L = lib()
        alive = np.ascontiguousarray(alive, dtype=np.uint8)
        begin = np.ascontiguousarray(begin, dtype=np.uint32)
        end = np.ascontiguousarray(end, dtype=np.uint32)
        n = int(L.synth_make_sensitive(self._h, alive.ctypes.data, begin.ctypes.data, end.ctypes.data))
        return Overlaps(**{f: _np(L.synth_field(self._h, 1, i), n, np.uint32) for i, f in enumerate(FIELDS)},
                        strand=_np(L.synth_strand(self._h, 1), n, np.uint8))

    def write_fasta(self, path):
        if lib().synth_write_fasta(self._h, path.encode()) != 0:
            raise IOError(path)

    def write_paf(self, path, sensitive=False, target_len=None):
        tl = None
        if target_len is not None:
            tl = np.ascontiguousarray(target_len, dtype=np.uint32)
        if lib().synth_write_paf(self._h, 1 if sensitive else 0, path.encode(),
                                 None if tl is None else tl.ctypes.data) != 0:
            raise IOError(path)

    def __del__(self):
        try:
            lib().synth_destroy(self._h)
        except Exception:
            pass
