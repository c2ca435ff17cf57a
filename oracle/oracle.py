"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes wrapper over oracle/_build/liboracle.so (flat restatement) and, where it has
been built, oracle/_ref/liboracle_ref.so (the same driver on the REAL rala::Pile /
rala::Overlap objects).  Only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FLAT = os.path.join(HERE, "_build", "liboracle.so")
REF = os.path.join(HERE, "_ref", "liboracle_ref.so")

KX, KA, KB, KAB, KBA = range(5)

_libs = {}

u32p = ctypes.c_void_p


def _load(path):
    L = ctypes.CDLL(path)
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    sig = {
        "ora_backend_name": (ctypes.c_char_p, []),
        "ora_create": (vp, [u32]),
        "ora_destroy": (None, [vp]),
        "ora_use_task_pool": (None, [vp]),
        "ora_set_reads": (None, [vp, vp, u64]),
        "ora_set_overlaps": (None, [vp, u64] + [vp] * 8),
        "ora_pass1": (None, [vp]),
        "ora_annotate": (i32, [vp]),
        "ora_initialize": (i32, [vp]),
        "ora_pass2": (None, [vp]),
        "ora_preprocess_chimeras": (None, [vp]),
        "ora_preprocess_repeats": (None, [vp, u64] + [vp] * 8),
        "ora_build_graph": (None, [vp]),
        "ora_remove_transitive_edges": (u32, [vp]),
        "ora_n_reads": (u64, [vp]),
        "ora_n_prefiltered": (u64, [vp]),
        "ora_get_valid": (None, [vp, vp]),
        "ora_get_piles": (None, [vp] + [vp] * 5),
        "ora_pile_data": (u64, [vp, u64, vp]),
        "ora_pile_row_digests": (None, [vp, vp, vp]),
        "ora_pile_intervals": (u64, [vp, u64, i32, vp]),
        "ora_pile_hill_counts": (u64, [vp, u64, vp]),
        "ora_pile_repeat_flags": (u64, [vp, u64, vp]),
        "ora_pile_set_state": (None, [vp, u64, vp, u32, u32, u32]),
        "ora_pile_add_layers": (None, [vp, u64, vp, u64]),
        "ora_pile_find_valid_region": (i32, [vp, u64]),
        "ora_pile_find_median": (None, [vp, u64]),
        "ora_pile_find_chimeric_hills": (None, [vp, u64]),
        "ora_pile_find_chimeric_pits": (None, [vp, u64]),
        "ora_pile_find_repetitive_hills": (None, [vp, u64, ctypes.c_uint16]),
        "ora_pile_to_json": (u64, [vp, u64, ctypes.c_char_p, u64]),
        "ora_pile_break_over_chimeric_pits": (i32, [vp, u64, ctypes.c_uint16]),
        "ora_pile_break_over_chimeric_hills": (i32, [vp, u64]),
        "ora_pile_find_slopes": (u64, [vp, u64, ctypes.c_double, vp, u64]),
        "ora_interval_merge": (u64, [vp, u64]),
        "ora_overlap_trim_type": (i32, [vp, u32, u32, u32, vp, vp]),
        "ora_get_overlaps": (u64, [vp, i32] + [vp] * 7),
        "ora_n_nodes": (u64, [vp]),
        "ora_n_edges": (u64, [vp]),
        "ora_get_nodes": (None, [vp, vp]),
        "ora_get_edges": (None, [vp] + [vp] * 4),
        "ora_set_graph": (None, [vp, u64, u64, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def build():
    subprocess.check_call(["make", "-s"], cwd=HERE)
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "ref"], cwd=HERE)


def have_ref():
    return os.path.exists(REF)


def lib(ref=False):
    key = "ref" if ref else "flat"
    if key not in _libs:
        path = REF if ref else FLAT
        if not os.path.exists(path) and not ref:
            build()
        _libs[key] = _load(path)
    return _libs[key]


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Oracle:
    """One run of the restated hot path.  ``ref=True`` uses the real reference objects."""

    def __init__(self, read_len, overlaps=None, n_threads=1, ref=False, task_pool=False):
        """task_pool: the per-pile fan-out as ONE TASK PER PILE through a thread pool with the interface of the reference's
        vendor/thread_pool (graph.cpp:367-377, 387-407) - the structure bench.py's cpu_baseline times"""
        self.L = lib(ref)
        self.h = self.L.ora_create(n_threads)
        if task_pool:
            self.L.ora_use_task_pool(self.h)
        self.read_len = _c(read_len, np.uint32)
        self.n_reads = int(self.read_len.shape[0])
        self.L.ora_set_reads(self.h, self.read_len.ctypes.data, self.n_reads)
        self.n_overlaps = 0
        if overlaps is not None:
            self.set_overlaps(overlaps)

    def __del__(self):
        try:
            self.L.ora_destroy(self.h)
        except Exception:
            pass

    @property
    def backend(self):
        return self.L.ora_backend_name().decode()

    def set_overlaps(self, ov):
        arrs = ov.arrays()
        self.n_overlaps = len(ov)
        self.L.ora_set_overlaps(self.h, self.n_overlaps, *[a.ctypes.data for a in arrs])

    # ---- stages ----
    def pass1(self):
        self.L.ora_pass1(self.h)

    def annotate(self):
        return self.L.ora_annotate(self.h)

    def initialize(self):
        return self.L.ora_initialize(self.h)

    def pass2(self):
        self.L.ora_pass2(self.h)

    def preprocess_chimeras(self):
        self.L.ora_preprocess_chimeras(self.h)

    def preprocess_repeats(self, sens):
        arrs = sens.arrays()
        self._sens_keep = arrs
        self.L.ora_preprocess_repeats(self.h, len(sens), *[a.ctypes.data for a in arrs])

    def build_graph(self):
        self.L.ora_build_graph(self.h)

    def remove_transitive_edges(self):
        return int(self.L.ora_remove_transitive_edges(self.h))

    def construct(self, sens=None):
        if self.initialize() != 0:
            return -1
        self.pass2()
        self.preprocess_chimeras()
        if sens is not None and len(sens):
            self.preprocess_repeats(sens)
        self.build_graph()
        return 0

    # ---- state ----
    def valid(self):
        out = np.zeros(self.n_overlaps, dtype=np.uint8)
        self.L.ora_get_valid(self.h, out.ctypes.data)
        return out

    def piles(self):
        n = self.n_reads
        d = dict(begin=np.zeros(n, np.uint32), end=np.zeros(n, np.uint32), median=np.zeros(n, np.uint16),
                 p10=np.zeros(n, np.uint16), alive=np.zeros(n, np.uint8))
        self.L.ora_get_piles(self.h, d["begin"].ctypes.data, d["end"].ctypes.data, d["median"].ctypes.data,
                             d["p10"].ctypes.data, d["alive"].ctypes.data)
        return d

    def pile_data(self, r):
        n = int(self.L.ora_pile_data(self.h, r, None))
        out = np.zeros(n, dtype=np.uint16)
        if n:
            self.L.ora_pile_data(self.h, r, out.ctypes.data)
        return out

    def pile_row_digests(self):
        """(fnv, sum) over every pile's data_ (FNV-1a-64 of its bytes; 0 for a read without a pile)"""
        fnv = np.zeros(self.n_reads, dtype=np.uint64)
        tot = np.zeros(self.n_reads, dtype=np.uint64)
        self.L.ora_pile_row_digests(self.h, fnv.ctypes.data, tot.ctypes.data)
        return fnv, tot

    def intervals(self, r, kind):
        n = int(self.L.ora_pile_intervals(self.h, r, kind, None))
        out = np.zeros((n, 2), dtype=np.uint32)
        if n:
            self.L.ora_pile_intervals(self.h, r, kind, out.ctypes.data)
        return out

    def all_intervals(self, kind):
        """CSR (offsets[n+1], flat[k,2]) of one interval kind over all reads."""
        offs = np.zeros(self.n_reads + 1, dtype=np.uint64)
        parts = []
        for r in range(self.n_reads):
            iv = self.intervals(r, kind)
            offs[r + 1] = offs[r] + len(iv)
            if len(iv):
                parts.append(iv)
        flat = np.concatenate(parts) if parts else np.zeros((0, 2), dtype=np.uint32)
        return offs, flat

    def hill_counts(self, r):
        n = int(self.L.ora_pile_hill_counts(self.h, r, None))
        out = np.zeros(n, dtype=np.uint32)
        if n:
            self.L.ora_pile_hill_counts(self.h, r, out.ctypes.data)
        return out

    def repeat_flags(self, r):
        n = int(self.L.ora_pile_repeat_flags(self.h, r, None))
        out = np.zeros(n, dtype=np.uint8)
        if n:
            self.L.ora_pile_repeat_flags(self.h, r, out.ctypes.data)
        return out

    # ---- unit-level ----
    def set_pile_state(self, r, data, begin, end):
        data = _c(data, np.uint16)
        self.L.ora_pile_set_state(self.h, r, data.ctypes.data, len(data), begin, end)

    def add_layers(self, r, bounds):
        b = _c(bounds, np.uint32)
        self.L.ora_pile_add_layers(self.h, r, b.ctypes.data, len(b))

    def find_valid_region(self, r):
        return bool(self.L.ora_pile_find_valid_region(self.h, r))

    def find_median(self, r):
        self.L.ora_pile_find_median(self.h, r)

    def find_chimeric_hills(self, r):
        self.L.ora_pile_find_chimeric_hills(self.h, r)

    def find_chimeric_pits(self, r):
        self.L.ora_pile_find_chimeric_pits(self.h, r)

    def find_repetitive_hills(self, r, med):
        self.L.ora_pile_find_repetitive_hills(self.h, r, med)

    def to_json(self, r):
        """Pile::to_json (reference pile.cpp:632-663) of a live read"""
        n = int(self.L.ora_pile_to_json(self.h, r, None, 0))
        buf = ctypes.create_string_buffer(n + 1)
        self.L.ora_pile_to_json(self.h, r, buf, n)
        return buf.raw[:n].decode()

    def break_over_chimeric_pits(self, r, med):
        return bool(self.L.ora_pile_break_over_chimeric_pits(self.h, r, med))

    def break_over_chimeric_hills(self, r):
        return bool(self.L.ora_pile_break_over_chimeric_hills(self.h, r))

    def find_slopes(self, r, q, cap=4096):
        out = np.zeros((cap, 2), dtype=np.uint32)
        n = int(self.L.ora_pile_find_slopes(self.h, r, q, out.ctypes.data, cap))
        if n > cap:
            return self.find_slopes(r, q, cap=n)
        return out[:n].copy()

    def interval_merge(self, iv):
        a = _c(iv, np.uint32).reshape(-1, 2).copy()
        n = int(self.L.ora_interval_merge(a.ctypes.data, len(a)))
        return a[:n].copy()

    def overlap_trim_type(self, a, b, strand, coords):
        c = _c(coords, np.uint32).copy()
        t = ctypes.c_int(-1)
        ok = self.L.ora_overlap_trim_type(self.h, a, b, strand, c.ctypes.data, ctypes.byref(t))
        return bool(ok), c, t.value

    def overlap_list(self, which=0):
        n = int(self.L.ora_get_overlaps(self.h, which, *([None] * 7)))
        d = dict(src=np.zeros(n, np.uint64), a_begin=np.zeros(n, np.uint32), a_end=np.zeros(n, np.uint32),
                 b_begin=np.zeros(n, np.uint32), b_end=np.zeros(n, np.uint32), length=np.zeros(n, np.uint32),
                 type=np.zeros(n, np.uint8))
        if n:
            self.L.ora_get_overlaps(self.h, which, *[d[k].ctypes.data for k in
                                                     ("src", "a_begin", "a_end", "b_begin", "b_end", "length",
                                                      "type")])
        return d

    # ---- graph ----
    def nodes(self):
        n = int(self.L.ora_n_nodes(self.h))
        out = np.zeros(n, dtype=np.uint32)
        if n:
            self.L.ora_get_nodes(self.h, out.ctypes.data)
        return out

    def edges(self):
        n = int(self.L.ora_n_edges(self.h))
        d = dict(src=np.zeros(n, np.uint32), dst=np.zeros(n, np.uint32), len=np.zeros(n, np.uint32),
                 marked=np.zeros(n, np.uint8))
        if n:
            self.L.ora_get_edges(self.h, d["src"].ctypes.data, d["dst"].ctypes.data, d["len"].ctypes.data,
                                 d["marked"].ctypes.data)
        return d

    def set_graph(self, n_nodes, src, dst, length):
        src, dst, length = _c(src, np.uint32), _c(dst, np.uint32), _c(length, np.uint32)
        self.L.ora_set_graph(self.h, n_nodes, len(src), src.ctypes.data, dst.ctypes.data, length.ctypes.data)
