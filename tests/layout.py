"""ctypes drivers of the two implementations of the clean-up stages after transitive reduction:
the product (rala_amd/host/libassembly_graph.so, index based) and the oracle restatement
(oracle/_build/liboracle.so, pointer based like the reference)."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = {"tips": 0, "bubbles": 1, "unitigs": 2, "shrink": 3, "long_edges": 4}
LAYOUT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                             ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32,
                             ctypes.c_double, ctypes.c_double, ctypes.c_double)


def numpy_engine(x, y, adj_off, adj, iterations, k, t, dt):
    """The layout steps of Graph::postprocess on the CPU, vectorised over the points but with
    every point's sums accumulated in ascending partner order (bit-identical to a sequential
    evaluation): the stand-in for rala_hip_layout in the CPU suite."""
    n = len(x)
    for _ in range(iterations):
        ax = np.zeros(n); ay = np.zeros(n)
        for m in range(n):
            dx = x - x[m]; dy = y - y[m]
            d = np.sqrt(dx * dx + dy * dy)
            d = np.where(d < 0.01, 0.01, d)
            s = (k * k) / (d * d)
            keep = np.arange(n) != m
            ax = np.where(keep, ax + dx * s, ax)
            ay = np.where(keep, ay + dy * s, ay)
        deg = np.diff(adj_off)
        for j in range(int(deg.max()) if n else 0):
            has = deg > j
            idx = np.where(has, adj[np.minimum(adj_off[:-1] + j, len(adj) - 1)] if len(adj) else 0, n)
            px = np.where(idx < n, x[np.minimum(idx, n - 1)], 0.0)
            py = np.where(idx < n, y[np.minimum(idx, n - 1)], 0.0)
            dx = x - px; dy = y - py
            d = np.sqrt(dx * dx + dy * dy)
            d = np.where(d < 0.01, 0.01, d)
            s = -1. * d / k
            ax = np.where(has, ax + dx * s, ax)
            ay = np.where(has, ay + dy * s, ay)
        length = np.sqrt(ax * ax + ay * ay)
        length = np.where(length < 0.01, 0.1, length)
        s = t / length
        x[:] = x + ax * s
        y[:] = y + ay * s
        t -= dt
    return 0


def engine_callback(fn):
    """wraps engine(x, y, adj_off, adj, iterations, k, t, dt) (numpy arrays, x / y in place)"""
    def raw(n, px, py, poff, padj, iterations, k, t, dt):
        x = np.ctypeslib.as_array(px, shape=(n,)); y = np.ctypeslib.as_array(py, shape=(n,))
        off = np.ctypeslib.as_array(poff, shape=(n + 1,)).copy()
        adj = np.ctypeslib.as_array(padj, shape=(int(off[n]),)).copy() if off[n] else np.zeros(0, np.uint32)
        return int(fn(x, y, off, adj, iterations, k, t, dt))
    return LAYOUT_FN(raw)
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def revcomp(s):
    return s.translate(_COMP)[::-1]


class _Graph:
    def __init__(self, lib, prefix):
        self.L, self.p = lib, prefix
        f = lambda name: getattr(lib, prefix + name)
        f("create").restype = ctypes.c_void_p
        f("add_node_pair").argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        f("add_edge").argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        f("mark_edge").argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        f("remove_marked").argtypes = [ctypes.c_void_p, ctypes.c_int]
        f("run").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
        f("run").restype = ctypes.c_uint32
        f("size").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        f("dump_nodes").argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 10
        f("dump_edges").argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
        f("destroy").argtypes = [ctypes.c_void_p]
        f("note_transitive").argtypes = [ctypes.c_void_p]
        f("edge_weights").argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        f("transitive").argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        f("transitive").restype = ctypes.c_uint64
        if prefix == "ag_":
            f("postprocess").argtypes = [ctypes.c_void_p, ctypes.c_uint32, LAYOUT_FN]
            f("postprocess").restype = ctypes.c_int
        else:
            f("postprocess").argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        f("node_data").argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64]
        f("node_data").restype = ctypes.c_uint64
        self.f = f
        self.h = f("create")()

    def __del__(self):
        if getattr(self, "h", None):
            self.f("destroy")(self.h)
            self.h = None

    def add_node_pair(self, seq_id, name, data):
        self.f("add_node_pair")(self.h, seq_id, name, data, revcomp(data))

    def add_edge(self, b, e, length):
        self.f("add_edge")(self.h, int(b), int(e), int(length))

    def mark_edge(self, e):
        self.f("mark_edge")(self.h, int(e))

    def remove_marked(self, remove_nodes=False):
        self.f("remove_marked")(self.h, int(remove_nodes))

    def note_transitive(self):
        self.f("note_transitive")(self.h)

    def postprocess(self, seed, engine=None):
        """force-directed layout -> edge weights; the product needs an engine for the layout steps"""
        if self.p == "ag_":
            cb = engine_callback(engine or numpy_engine)
            assert self.f("postprocess")(self.h, seed, cb) == 0
        else:
            self.f("postprocess")(self.h, seed)

    def edge_weights(self):
        ne = self.dump()[1]["alive"].shape[0]
        w = np.zeros(ne, dtype=np.float64)
        self.f("edge_weights")(self.h, w.ctypes.data)
        return w

    def transitive(self):
        n = int(self.f("transitive")(self.h, None))
        out = np.zeros((n, 2), dtype=np.uint64)
        if n:
            self.f("transitive")(self.h, out.ctypes.data)
        return out

    def run(self, op, arg=0):
        return int(self.f("run")(self.h, OPS[op], arg))

    def print(self, kind):
        """the graph in one of the reference's on-disk formats: 'csv', 'gfa', 'json' (piles as stand-ins)"""
        k = {"csv": 0, "gfa": 1, "json": 2}[kind]
        fn = self.f("print")
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64]
        fn.restype = ctypes.c_uint64
        n = int(fn(self.h, k, None, 0))
        buf = ctypes.create_string_buffer(n + 1)
        fn(self.h, k, buf, n)
        return buf.raw[:n]

    def node_data(self, node):
        n = int(self.f("node_data")(self.h, node, None, 0))
        buf = ctypes.create_string_buffer(n + 1)
        self.f("node_data")(self.h, node, buf, n)
        return buf.raw[:n]

    def dump(self):
        nn, ne = ctypes.c_uint64(), ctypes.c_uint64()
        self.f("size")(self.h, ctypes.byref(nn), ctypes.byref(ne))
        nn, ne = nn.value, ne.value
        spec = [("alive", np.uint8), ("length", np.uint32), ("n_seq", np.uint32), ("data_hash", np.uint64),
                ("ids_hash", np.uint64), ("first_rc", np.uint8), ("last_rc", np.uint8), ("indeg", np.uint32),
                ("outdeg", np.uint32), ("adj_hash", np.uint64)]
        nodes = {k: np.zeros(nn, dtype=t) for k, t in spec}
        self.f("dump_nodes")(self.h, *[nodes[k].ctypes.data for k, _ in spec])
        espec = [("alive", np.uint8), ("begin", np.uint32), ("end", np.uint32), ("length", np.uint32)]
        edges = {k: np.zeros(ne, dtype=t) for k, t in espec}
        self.f("dump_edges")(self.h, *[edges[k].ctypes.data for k, _ in espec])
        # hashes of dead nodes are not comparable (the product clears them, the oracle frees them)
        dead = nodes["alive"] == 0
        for k in ("length", "n_seq", "data_hash", "ids_hash", "indeg", "outdeg", "adj_hash"):
            nodes[k][dead] = 0
        return nodes, edges


def product():
    return _Graph(ctypes.CDLL(os.path.join(ROOT, "rala_amd", "host", "libassembly_graph.so")), "ag_")


def oracle():
    return _Graph(ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so")), "ol_")


def assert_same_graph(a, b, what=""):
    (na, ea), (nb, eb) = a.dump(), b.dump()
    wa, wb = a.edge_weights(), b.edge_weights()
    assert wa.shape == wb.shape and (wa == wb).all(), "%s edge weights differ (max %g)" % (what, np.abs(wa - wb).max())
    ta, tb = a.transitive(), b.transitive()
    assert ta.shape == tb.shape and (ta == tb).all(), "%s transitive-edge lists differ" % what
    for name, x, y in (("node", na, nb), ("edge", ea, eb)):
        for k in x:
            assert x[k].shape == y[k].shape, "%s %s.%s: %s vs %s" % (what, name, k, x[k].shape, y[k].shape)
            if not (x[k] == y[k]).all():
                i = int(np.nonzero(x[k] != y[k])[0][0])
                raise AssertionError("%s %s.%s differs first at %d: product %s oracle %s" % (what, name, k, i, x[k][i], y[k][i]))
