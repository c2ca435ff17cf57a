"""ctypes binding of librala_hip's C ABI (include/rala_hip.h).

Used by the tests and by bench.py.  There is no fallback of any kind: if the shared
library is missing or no HIP device is usable, construction raises.
"""
import ctypes
import os

import numpy as np

from . import build as _build

TYPE_X, TYPE_A, TYPE_B, TYPE_AB, TYPE_BA = range(5)
NO_READ = 0xFFFFFFFF
MEM_HOST, MEM_DEVICE, MEM_HOST_ASYNC = 0, 1, 2

ERRORS = {0: "OK", -1: "EDEVICE", -2: "EINVAL", -3: "ECAPACITY", -4: "EFILTERED", -5: "ENOMEM", -6: "ENOTAFILE", -7: "ETOOLARGE"}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "librala_hip.so")

# every symbol include/rala_hip.h declares
SYMBOLS = (
    "rala_hip_create", "rala_hip_destroy", "rala_hip_last_error", "rala_hip_set_option", "rala_hip_stream",
    "rala_hip_set_reads", "rala_hip_set_overlaps", "rala_hip_initialize", "rala_hip_construct",
    "rala_hip_remove_transitive_edges", "rala_hip_tr_mark", "rala_hip_get_valid", "rala_hip_get_piles",
    "rala_hip_get_pile_data", "rala_hip_get_pile_row_digests", "rala_hip_get_intervals", "rala_hip_get_overlaps", "rala_hip_get_graph_size",
    "rala_hip_get_graph", "rala_hip_get_timings", "rala_hip_get_num_prefiltered",
    "rala_hip_dedupe", "rala_hip_emit_bound_tuples", "rala_hip_set_bound_tuples", "rala_hip_import_state",
    "rala_hip_emit_bound_tuples_bucketed", "rala_hip_get_device_state", "rala_hip_import_state_device",
    "rala_hip_bound_records_fit", "rala_hip_emit_bound_records_bucketed", "rala_hip_set_bound_records",
    "rala_hip_copy_device_state", "rala_hip_layout", "rala_hip_find_repetitive_hills",
    "rala_hip_mg_unique_id", "rala_hip_mg_local_group_create", "rala_hip_mg_local_group_destroy", "rala_hip_mg_create",
    "rala_hip_mg_create_contexts", "rala_hip_mg_join", "rala_hip_set_name_table", "rala_hip_set_overlaps_from_paf", "rala_hip_set_overlaps_from_mhap",
    "rala_hip_get_ingest_timings", "rala_hip_get_overlap_columns", "rala_hip_tokenise_sensitive_paf",
    "rala_hip_mg_set_overlaps_from_paf", "rala_hip_mg_get_slice",
    "rala_hip_mg_destroy", "rala_hip_mg_last_error", "rala_hip_mg_set_reads", "rala_hip_mg_slice_cuts",
    "rala_hip_mg_set_overlaps", "rala_hip_mg_run", "rala_hip_mg_run_threads", "rala_hip_mg_context",
    "rala_hip_mg_owner_context",
    "rala_hip_mg_get_pile_data", "rala_hip_mg_get_pile_row_digests", "rala_hip_mg_get_timings",
)
COMM_RCCL, COMM_LOCAL = 0, 1


class OverlapsC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("a_id", "b_id", "a_begin", "a_end", "b_begin", "b_end", "length", "strand")]


class DeviceState(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("begin", "end", "median", "p10", "alive", "n_pits", "n_hills", "slot",
                                               "pool")] + [("pool_count", ctypes.c_uint64), ("valid", ctypes.c_void_p)]


class Timings(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in
                ("dedupe_ms", "bucket_ms", "pile_ms", "classify_ms", "death_ms", "finish_ms", "tail_host_ms",
                 "tr_ms", "total_ms")] + [("pile_launches", ctypes.c_uint32), ("death_rounds", ctypes.c_uint32),
                                  ("pile_overflow_reads", ctypes.c_uint32),
                                  ("pile_position_reads", ctypes.c_uint32),
                                  ("pile_unbounded_reads", ctypes.c_uint32), ("pool_regrown", ctypes.c_uint32),
                                  ("repeats_ms", ctypes.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MgTimings(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("emit_ms", "exchange_ms", "owner_ms", "gather_ms", "construct_ms",
                                              "repeats_ms", "tr_ms", "total_ms")] + [("tuples_sent", ctypes.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def lib(build=True):
    """Load librala_hip.so (building it in-tree first when sources are newer)."""
    global _lib
    if _lib is None:
        path = LIB_PATH
        if build:
            path = _build.build_hip()
        if os.environ.get("RALA_HIP_LIB_AB"):        # (measurements: another build of the library, tools/gpurun/r6_build_ab.sh)
            path = os.environ["RALA_HIP_LIB_AB"]
        if not os.path.exists(path):
            raise RuntimeError("librala_hip.so is missing: run __graft_entry__.build()")
        L = ctypes.CDLL(path)
        vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
        L.rala_hip_create.argtypes = [i32, ctypes.POINTER(vp)]
        L.rala_hip_destroy.argtypes = [vp]
        L.rala_hip_destroy.restype = None
        L.rala_hip_last_error.argtypes = [vp]
        L.rala_hip_last_error.restype = ctypes.c_char_p
        L.rala_hip_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
        L.rala_hip_stream.argtypes = [vp]
        L.rala_hip_stream.restype = vp
        L.rala_hip_set_reads.argtypes = [vp, vp, u64]
        L.rala_hip_set_overlaps.argtypes = [vp, ctypes.POINTER(OverlapsC), u64, i32]
        L.rala_hip_initialize.argtypes = [vp]
        L.rala_hip_construct.argtypes = [vp, ctypes.POINTER(OverlapsC), u64]
        L.rala_hip_remove_transitive_edges.argtypes = [vp, ctypes.POINTER(u32)]
        L.rala_hip_tr_mark.argtypes = [vp, u32, u32, vp, vp, vp, vp, ctypes.POINTER(u32)]
        L.rala_hip_get_valid.argtypes = [vp, vp]
        L.rala_hip_get_piles.argtypes = [vp, vp, vp, vp, vp, vp]
        L.rala_hip_get_pile_data.argtypes = [vp, u64, vp]
        L.rala_hip_get_pile_row_digests.argtypes = [vp, vp, vp, vp]
        L.rala_hip_get_intervals.argtypes = [vp, i32, vp, vp, vp]
        L.rala_hip_get_overlaps.argtypes = [vp, i32, ctypes.POINTER(u64), vp, vp, vp, vp, vp, vp, vp]
        L.rala_hip_get_graph_size.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
        L.rala_hip_get_graph.argtypes = [vp, vp, vp, vp, vp, vp]
        L.rala_hip_get_timings.argtypes = [vp, ctypes.POINTER(Timings)]
        L.rala_hip_get_num_prefiltered.argtypes = [vp, ctypes.POINTER(u64)]
        L.rala_hip_dedupe.argtypes = [vp]
        L.rala_hip_emit_bound_tuples.argtypes = [vp, vp]
        L.rala_hip_set_bound_tuples.argtypes = [vp, vp, u64, i32]
        L.rala_hip_import_state.argtypes = [vp] + [vp] * 11
        L.rala_hip_emit_bound_tuples_bucketed.argtypes = [vp, u32, vp, vp]
        L.rala_hip_get_device_state.argtypes = [vp, ctypes.POINTER(DeviceState)]
        L.rala_hip_import_state_device.argtypes = [vp, ctypes.POINTER(DeviceState)]
        L.rala_hip_copy_device_state.argtypes = [vp, ctypes.POINTER(DeviceState)]
        L.rala_hip_layout.argtypes = [vp, u32, vp, vp, vp, vp, u32, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.rala_hip_find_repetitive_hills.argtypes = [vp, u64, u32, u32, ctypes.c_uint16, ctypes.c_uint16, ctypes.c_uint16]
        L.rala_hip_mg_unique_id.argtypes = [vp]
        L.rala_hip_mg_local_group_create.argtypes = [u32, ctypes.POINTER(vp)]
        L.rala_hip_mg_local_group_destroy.argtypes = [vp]
        L.rala_hip_mg_local_group_destroy.restype = None
        L.rala_hip_mg_create.argtypes = [i32, u32, u32, i32, vp, ctypes.POINTER(vp)]
        L.rala_hip_mg_create_contexts.argtypes = [i32, u32, u32, ctypes.POINTER(vp)]
        L.rala_hip_mg_join.argtypes = [vp, i32, vp]
        L.rala_hip_mg_destroy.argtypes = [vp]
        L.rala_hip_mg_destroy.restype = None
        L.rala_hip_mg_last_error.argtypes = [vp]
        L.rala_hip_mg_last_error.restype = ctypes.c_char_p
        L.rala_hip_mg_set_reads.argtypes = [vp, vp, u64]
        L.rala_hip_mg_slice_cuts.argtypes = [vp, vp, u64, u32, vp]
        L.rala_hip_mg_set_overlaps.argtypes = [vp, ctypes.POINTER(OverlapsC), u64, u64, i32]
        L.rala_hip_mg_run.argtypes = [vp, ctypes.POINTER(OverlapsC), u64, ctypes.POINTER(u32)]
        L.rala_hip_mg_run_threads.argtypes = [vp, u32, vp, vp, ctypes.POINTER(u32)]
        L.rala_hip_mg_context.argtypes = [vp]
        L.rala_hip_mg_context.restype = vp
        L.rala_hip_mg_owner_context.argtypes = [vp]
        L.rala_hip_mg_owner_context.restype = vp
        L.rala_hip_mg_get_pile_data.argtypes = [vp, u64, vp]
        L.rala_hip_mg_get_pile_row_digests.argtypes = [vp, vp, vp, vp]
        L.rala_hip_mg_get_timings.argtypes = [vp, ctypes.POINTER(MgTimings)]
        _lib = L
    return _lib


class RalaHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("librala_hip: %s (%d): %s" % (ERRORS.get(code, "?"), code, text))
        self.code = code


class DeviceOverlaps:
    """overlap columns that lie in device memory (kept alive by the caller, e.g. torch tensors): name -> pointer.
    Accepted where a sensitive set is (Context.construct, ShardedRank.run, run_ranks) once the context's option
    sensitive_in_device_memory is 1."""

    def __init__(self, ptrs, n, keep=None):
        self.ptrs, self.n, self.keep = dict(ptrs), int(n), keep

    def __len__(self):
        return self.n

    @classmethod
    def from_host(cls, ov, device):
        """upload numpy columns to `device` (the HIP runtime directly: hipMalloc / hipMemcpy, freed with the object)"""
        rt = _hip_runtime()
        keep, ptrs = _DeviceBlocks(rt), {}
        _rt_check(rt.hipSetDevice(int(device)), "hipSetDevice")
        for name, _ in OverlapsC._fields_:
            arr = np.ascontiguousarray(getattr(ov, name))
            p = ctypes.c_void_p()
            _rt_check(rt.hipMalloc(ctypes.byref(p), max(1, arr.nbytes)), "hipMalloc")
            keep.blocks.append(p)
            if arr.nbytes:
                _rt_check(rt.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1), "hipMemcpy")        # hipMemcpyHostToDevice
            ptrs[name] = p.value
        return cls(ptrs, len(ov), keep)


_rt = None


def _hip_runtime():
    global _rt
    if _rt is None:
        lib()                           # (librala_hip.so has brought the runtime into the process)
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _rt = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _rt is None:
            raise RalaHipError(-2, "libamdhip64 not found")
        _rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _rt.hipFree.argtypes = [ctypes.c_void_p]
        _rt.hipSetDevice.argtypes = [ctypes.c_int]
    return _rt


def _rt_check(rc, what):
    if rc != 0:
        raise RalaHipError(-2, "%s failed (%d)" % (what, rc))


class _DeviceBlocks:
    def __init__(self, rt):
        self.rt, self.blocks = rt, []

    def __del__(self):
        try:
            for p in self.blocks:
                self.rt.hipFree(p)
        except Exception:
            pass


def _soa(ov):
    """OverlapsC over an object with numpy members a_id … strand (kept alive by the caller), or over DeviceOverlaps."""
    c = OverlapsC()
    if isinstance(ov, DeviceOverlaps):
        for name, _ in OverlapsC._fields_:
            setattr(c, name, ov.ptrs[name])
        return c
    for name, _ in OverlapsC._fields_:
        arr = getattr(ov, name)
        want = np.uint8 if name == "strand" else np.uint32
        assert arr.dtype == want and arr.flags["C_CONTIGUOUS"], name
        setattr(c, name, arr.ctypes.data)
    return c


class Context:
    """One librala_hip context = one data set on one GPU."""

    def __init__(self, device=0, _borrowed=None):
        self.L = lib()
        self._owned = _borrowed is None
        if _borrowed is not None:
            h = ctypes.c_void_p(_borrowed)
        else:
            h = ctypes.c_void_p()
            rc = self.L.rala_hip_create(device, ctypes.byref(h))
            if rc != 0:
                raise RalaHipError(rc, "no usable HIP device %d" % device)
        self.h = h
        self.n_reads = 0
        self.n_overlaps = 0
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                self.L.rala_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RalaHipError(rc, self.L.rala_hip_last_error(self.h).decode())

    def set_option(self, key, value):
        self._check(self.L.rala_hip_set_option(self.h, key.encode(), int(value)))

    def stream(self):
        return self.L.rala_hip_stream(self.h)

    # ---- inputs ----
    def set_reads(self, read_len):
        rl = np.ascontiguousarray(read_len, dtype=np.uint32)
        self.read_len = rl
        self.n_reads = int(rl.shape[0])
        self._check(self.L.rala_hip_set_reads(self.h, rl.ctypes.data, self.n_reads))

    def set_overlaps(self, ov, later=False):
        """later: RALA_HIP_MEM_HOST_ASYNC - the columns (page-locked, valid until initialize() has returned) are uploaded by
        initialize(), beside its first kernels"""
        c = _soa(ov)
        self.n_overlaps = len(ov)
        self._keep_columns = ov if later else None
        self._check(self.L.rala_hip_set_overlaps(self.h, ctypes.byref(c), self.n_overlaps, MEM_HOST_ASYNC if later else MEM_HOST))

    def set_overlaps_device(self, ptrs, n):
        """ptrs: dict name -> device pointer (int); the memory must outlive the context's use."""
        c = OverlapsC()
        for name, _ in OverlapsC._fields_:
            setattr(c, name, ptrs[name])
        self.n_overlaps = int(n)
        self._check(self.L.rala_hip_set_overlaps(self.h, ctypes.byref(c), self.n_overlaps, MEM_DEVICE))

    # ---- multi-GPU building blocks ----
    def dedupe(self):
        self._check(self.L.rala_hip_dedupe(self.h))

    def emit_bound_tuples(self, tuples_ptr):
        """device pointer of 4 * n_overlaps 8-byte tuples (read | bound << 32), 16-byte aligned"""
        self._check(self.L.rala_hip_emit_bound_tuples(self.h, tuples_ptr))

    def layout(self, x, y, adj_off, adj, iterations, k, t, dt):
        """force-directed layout steps (rala_hip_layout); x, y float64 arrays updated in place"""
        assert x.dtype == np.float64 and y.dtype == np.float64 and x.flags.c_contiguous and y.flags.c_contiguous
        adj_off = np.ascontiguousarray(adj_off, dtype=np.uint32)
        adj = np.ascontiguousarray(adj, dtype=np.uint32)
        self._check(self.L.rala_hip_layout(self.h, len(x), x.ctypes.data, y.ctypes.data, adj_off.ctypes.data,
                                           adj.ctypes.data if len(adj) else None, iterations, k, t, dt))

    def emit_bound_tuples_bucketed(self, world, tuples_ptr):
        """tuples grouped by owner rank; returns the bucket sizes"""
        counts = np.zeros(world, dtype=np.uint64)
        self._check(self.L.rala_hip_emit_bound_tuples_bucketed(self.h, world, tuples_ptr, counts.ctypes.data))
        return counts

    def device_state(self):
        st = DeviceState()
        self._check(self.L.rala_hip_get_device_state(self.h, ctypes.byref(st)))
        return st

    def copy_device_state(self, **ptrs):
        """device-to-device copy of the named arrays into caller buffers (pointers)"""
        st = DeviceState()
        for k, v in ptrs.items():
            setattr(st, k, v)
        self._check(self.L.rala_hip_copy_device_state(self.h, ctypes.byref(st)))

    def import_state_device(self, **ptrs):
        """ptrs: begin, end, median, p10, alive, n_pits, n_hills, slot, pool, pool_count, valid"""
        st = DeviceState()
        for k, v in ptrs.items():
            setattr(st, k, v)
        self._check(self.L.rala_hip_import_state_device(self.h, ctypes.byref(st)))

    def set_bound_tuples_device(self, tuples_ptr, n):
        self.n_overlaps = 0
        self._check(self.L.rala_hip_set_bound_tuples(self.h, tuples_ptr, int(n), MEM_DEVICE))

    def set_bound_tuples(self, reads, bounds):
        """host arrays (local read, bound) -> packed 8-byte tuples"""
        t = np.ascontiguousarray(reads, dtype=np.uint64) | (np.ascontiguousarray(bounds, dtype=np.uint64) << np.uint64(32))
        self.n_overlaps = 0
        self._check(self.L.rala_hip_set_bound_tuples(self.h, t.ctypes.data, len(t), MEM_HOST))

    def import_state(self, valid, piles, pits, hills):
        """piles: dict as returned by piles(); pits / hills: (offsets, pairs, aux) as intervals()."""
        c = np.ascontiguousarray
        valid = c(valid, dtype=np.uint8)
        arrs = [valid, c(piles["begin"], dtype=np.uint32), c(piles["end"], dtype=np.uint32),
                c(piles["median"], dtype=np.uint16), c(piles["p10"], dtype=np.uint16),
                c(piles["alive"], dtype=np.uint8), c(pits[0], dtype=np.uint64),
                c(pits[1], dtype=np.uint32).reshape(-1), c(pits[2], dtype=np.uint32),
                c(hills[0], dtype=np.uint64), c(hills[1], dtype=np.uint32).reshape(-1)]
        ptrs = [a.ctypes.data if a.size else None for a in arrs]
        if ptrs[6] is None or ptrs[9] is None:
            raise ValueError("interval offsets must have n_reads + 1 entries")
        self._check(self.L.rala_hip_import_state(self.h, *ptrs))

    # ---- stages ----
    def initialize(self):
        self._check(self.L.rala_hip_initialize(self.h))

    def construct(self, sens=None):
        if sens is not None and len(sens):
            c = _soa(sens)
            self._check(self.L.rala_hip_construct(self.h, ctypes.byref(c), len(sens)))
        else:
            self._check(self.L.rala_hip_construct(self.h, None, 0))

    def find_repetitive_hills(self, read, begin, end, median, p10, dataset_median):
        self._check(self.L.rala_hip_find_repetitive_hills(self.h, int(read), int(begin), int(end), int(median), int(p10),
                                                          int(dataset_median)))

    def remove_transitive_edges(self):
        n = ctypes.c_uint32(0)
        self._check(self.L.rala_hip_remove_transitive_edges(self.h, ctypes.byref(n)))
        return int(n.value)

    def tr_mark(self, n_nodes, src, dst, length):
        src = np.ascontiguousarray(src, dtype=np.uint32)
        dst = np.ascontiguousarray(dst, dtype=np.uint32)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        marks = np.zeros(len(src), dtype=np.uint8)
        n = ctypes.c_uint32(0)
        self._check(self.L.rala_hip_tr_mark(self.h, n_nodes, len(src), src.ctypes.data, dst.ctypes.data,
                                            length.ctypes.data, marks.ctypes.data, ctypes.byref(n)))
        return marks, int(n.value)

    # ---- results ----
    def valid(self):
        out = np.zeros(self.n_overlaps, dtype=np.uint8)
        self._check(self.L.rala_hip_get_valid(self.h, out.ctypes.data))
        return out

    def piles(self):
        n = self.n_reads
        d = dict(begin=np.zeros(n, np.uint32), end=np.zeros(n, np.uint32), median=np.zeros(n, np.uint16),
                 p10=np.zeros(n, np.uint16), alive=np.zeros(n, np.uint8))
        self._check(self.L.rala_hip_get_piles(self.h, d["begin"].ctypes.data, d["end"].ctypes.data,
                                              d["median"].ctypes.data, d["p10"].ctypes.data,
                                              d["alive"].ctypes.data))
        return d

    def pile_data(self, r):
        out = np.zeros(int(self.read_len[r]), dtype=np.uint16)
        self._check(self.L.rala_hip_get_pile_data(self.h, r, out.ctypes.data))
        return out

    def pile_row_digests(self):
        """checksums of every pile row, computed where the rows lie: (fnv, inside, outside), uint64 per read -
        FNV-1a-64 of Pile::data(), the row's sum inside the valid region, the stored values' sum outside it"""
        out = [np.zeros(self.n_reads, dtype=np.uint64) for _ in range(3)]
        self._check(self.L.rala_hip_get_pile_row_digests(self.h, *[a.ctypes.data for a in out]))
        return tuple(out)

    def intervals(self, kind):
        """(offsets[n+1] uint64, pairs[k,2] uint32, aux[k] uint32)"""
        offs = np.zeros(self.n_reads + 1, dtype=np.uint64)
        self._check(self.L.rala_hip_get_intervals(self.h, kind, offs.ctypes.data, None, None))
        k = int(offs[-1])
        pairs = np.zeros((k, 2), dtype=np.uint32)
        aux = np.zeros(k, dtype=np.uint32)
        if k:
            self._check(self.L.rala_hip_get_intervals(self.h, kind, offs.ctypes.data, pairs.ctypes.data,
                                                      aux.ctypes.data))
        return offs, pairs, aux

    def overlap_list(self, which=0):
        n = ctypes.c_uint64(0)
        self._check(self.L.rala_hip_get_overlaps(self.h, which, ctypes.byref(n), *([None] * 7)))
        m = int(n.value)
        d = dict(src=np.zeros(m, np.uint32), a_begin=np.zeros(m, np.uint32), a_end=np.zeros(m, np.uint32),
                 b_begin=np.zeros(m, np.uint32), b_end=np.zeros(m, np.uint32), length=np.zeros(m, np.uint32),
                 type=np.zeros(m, np.uint8))
        if m:
            self._check(self.L.rala_hip_get_overlaps(self.h, which, ctypes.byref(n), *[
                d[k].ctypes.data for k in ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")]))
        return d

    def graph(self):
        nn, ne = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.L.rala_hip_get_graph_size(self.h, ctypes.byref(nn), ctypes.byref(ne)))
        nn, ne = int(nn.value), int(ne.value)
        d = dict(node_read=np.zeros(nn, np.uint32), src=np.zeros(ne, np.uint32), dst=np.zeros(ne, np.uint32),
                 len=np.zeros(ne, np.uint32), marked=np.zeros(ne, np.uint8))
        self._check(self.L.rala_hip_get_graph(self.h, d["node_read"].ctypes.data, d["src"].ctypes.data,
                                              d["dst"].ctypes.data, d["len"].ctypes.data,
                                              d["marked"].ctypes.data))
        return d

    def timings(self):
        t = Timings()
        self._check(self.L.rala_hip_get_timings(self.h, ctypes.byref(t)))
        return t.as_dict()

    def num_prefiltered(self):
        n = ctypes.c_uint64(0)
        self._check(self.L.rala_hip_get_num_prefiltered(self.h, ctypes.byref(n)))
        return int(n.value)


def slice_cuts(a_id, world, b_id=None):
    """rala_hip_mg_slice_cuts: world + 1 file positions, cuts on a_id-run boundaries (records whose
    query or target is NO_READ do not break a run)"""
    a = np.ascontiguousarray(a_id, dtype=np.uint32)
    b = None if b_id is None else np.ascontiguousarray(b_id, dtype=np.uint32)
    cuts = np.zeros(world + 1, dtype=np.uint64)
    rc = lib().rala_hip_mg_slice_cuts(a.ctypes.data if len(a) else None, b.ctypes.data if b is not None and len(b) else None,
                                      len(a), world, cuts.ctypes.data)
    if rc != 0:
        raise RalaHipError(rc, "slice_cuts")
    return [int(c) for c in cuts]


def unique_id():
    """128-byte RCCL id (rank 0 makes it, the others receive the bytes)"""
    buf = ctypes.create_string_buffer(128)
    rc = lib().rala_hip_mg_unique_id(buf)
    if rc != 0:
        raise RalaHipError(rc, "no usable RCCL")
    return buf.raw


class ShardedRank:
    """One rank of a sharded run (rala_hip_mg_*).  token: 128-byte RCCL id (bytes) or a LocalGroup; token None: only the
    rank's device contexts are created (not collective) and join(token) enters the group later - launchers make sure
    every rank has its contexts before anybody joins."""

    def __init__(self, device, rank, world, token=None):
        self.L = lib()
        self.rank, self.world = rank, world
        h = ctypes.c_void_p()
        rc = self.L.rala_hip_mg_create_contexts(device, rank, world, ctypes.byref(h))
        if rc != 0:
            raise RalaHipError(rc, "rala_hip_mg_create_contexts failed (device %d, rank %d of %d)" % (device, rank, world))
        self.h = h
        self._keep = None
        self._token = None
        if token is not None:
            try:
                self.join(token)
            except Exception:
                self.close()
                raise

    def join(self, token):
        """collective: every rank of the group calls it (ncclCommInitRank / the in-process rendezvous)"""
        if isinstance(token, LocalGroup):
            self._token = token
            rc = self.L.rala_hip_mg_join(self.h, COMM_LOCAL, token.h)
        else:
            self._token = ctypes.create_string_buffer(bytes(token), 128)
            rc = self.L.rala_hip_mg_join(self.h, COMM_RCCL, self._token)
        self._check(rc)

    def close(self):
        if getattr(self, "h", None):
            self.L.rala_hip_mg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RalaHipError(rc, self.L.rala_hip_mg_last_error(self.h).decode())

    def set_reads(self, read_len):
        rl = np.ascontiguousarray(read_len, dtype=np.uint32)
        self.read_len = rl
        self._check(self.L.rala_hip_mg_set_reads(self.h, rl.ctypes.data, len(rl)))

    def set_overlaps(self, ov_slice, first):
        c = _soa(ov_slice)
        self.n_slice = len(ov_slice)
        self._check(self.L.rala_hip_mg_set_overlaps(self.h, ctypes.byref(c), len(ov_slice), int(first), MEM_HOST))

    def run(self, sens_slice=None):
        n = ctypes.c_uint32(0)
        if sens_slice is not None:          # an EMPTY share still takes part in the collective sensitive pass
            c = _soa(sens_slice)
            self._check(self.L.rala_hip_mg_run(self.h, ctypes.byref(c), len(sens_slice), ctypes.byref(n)))
        else:
            self._check(self.L.rala_hip_mg_run(self.h, None, 0, ctypes.byref(n)))
        return int(n.value)

    def context(self):
        """the replicated result as a (borrowed) Context: piles(), intervals(), overlap_list(), graph() ..."""
        c = Context(_borrowed=self.L.rala_hip_mg_context(self.h))
        c.read_len = self.read_len
        c.n_reads = len(self.read_len)
        c.n_overlaps = self.n_slice
        return c

    def pile_data(self, r):
        out = np.zeros(int(self.read_len[r]), dtype=np.uint16)
        self._check(self.L.rala_hip_mg_get_pile_data(self.h, int(r), out.ctypes.data))
        return out

    def pile_row_digests(self):
        """(fnv, inside, outside) of the rows this rank owns: entry j = read j * world + rank"""
        n_own = len(range(self.rank, len(self.read_len), self.world))
        out = [np.zeros(n_own, dtype=np.uint64) for _ in range(3)]
        self._check(self.L.rala_hip_mg_get_pile_row_digests(self.h, *[a.ctypes.data for a in out]))
        return tuple(out)

    def owner_timings(self):
        """stage timings of the owner context (bucketing and pile kernels over the owned reads)"""
        c = Context(_borrowed=self.L.rala_hip_mg_owner_context(self.h))
        return c.timings()

    def timings(self):
        t = MgTimings()
        self._check(self.L.rala_hip_mg_get_timings(self.h, ctypes.byref(t)))
        return t.as_dict()


class LocalGroup:
    """rendezvous object of ranks that are threads of this process (RALA_HIP_COMM_LOCAL)"""

    def __init__(self, world):
        self.L = lib()
        h = ctypes.c_void_p()
        rc = self.L.rala_hip_mg_local_group_create(world, ctypes.byref(h))
        if rc != 0:
            raise RalaHipError(rc, "local group")
        self.h, self.world = h, world

    def close(self):
        if getattr(self, "h", None):
            self.L.rala_hip_mg_local_group_destroy(self.h)
            self.h = None


def run_ranks(ranks, sens_slices=None):
    """rala_hip_mg_run_threads: all ranks of this process, one host thread each"""
    L = lib()
    n = len(ranks)
    arr = (ctypes.c_void_p * n)(*[r.h for r in ranks])
    pairs = ctypes.c_uint32(0)
    if sens_slices is not None and any(s is None for s in sens_slices):
        # (tests) ranks that disagree on whether there is a sensitive pass: one Python thread per rank
        import threading
        rcs = [0] * n

        def one(k):
            try:
                ranks[k].run(sens_slices[k])
            except RalaHipError as e:
                rcs[k] = e
        th = [threading.Thread(target=one, args=(k,)) for k in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        bad = [e for e in rcs if e != 0]
        if bad:
            raise bad[0]
        return 0
    if sens_slices is not None:
        cs = (OverlapsC * n)(*[_soa(s) for s in sens_slices])
        ns = (ctypes.c_uint64 * n)(*[len(s) for s in sens_slices])
        rc = L.rala_hip_mg_run_threads(arr, n, cs, ns, ctypes.byref(pairs))
    else:
        rc = L.rala_hip_mg_run_threads(arr, n, None, None, ctypes.byref(pairs))
    if rc != 0:
        msgs = [L.rala_hip_mg_last_error(r.h).decode() for r in ranks]
        raise RalaHipError(rc, " | ".join(m for m in msgs if m))
    return int(pairs.value)
