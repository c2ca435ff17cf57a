"""ctypes drivers of the two implementations of the clean-up stages after transitive reduction:
the product (rala_amd/host/libassembly_graph.so, index based) and the oracle restatement
(oracle/_build/liboracle.so, pointer based like the reference)."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = {"tips": 0, "bubbles": 1, "unitigs": 2, "shrink": 3, "long_edges": 4}
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

    def run(self, op, arg=0):
        return int(self.f("run")(self.h, OPS[op], arg))

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
    for name, x, y in (("node", na, nb), ("edge", ea, eb)):
        for k in x:
            assert x[k].shape == y[k].shape, "%s %s.%s: %s vs %s" % (what, name, k, x[k].shape, y[k].shape)
            if not (x[k] == y[k]).all():
                i = int(np.nonzero(x[k] != y[k])[0][0])
                raise AssertionError("%s %s.%s differs first at %d: product %s oracle %s" % (what, name, k, i, x[k][i], y[k][i]))
