"""CPU: the multi-threaded PAF reader gives the same columns as the
line-by-line reader, for every thread count, on synthetic files and on awkward ones."""
import ctypes
import os

import numpy as np
import pytest

from rala_amd.synth import Dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("a_id", "b_id", "a_begin", "a_end", "b_begin", "b_end", "length")


def _lib():
    L = ctypes.CDLL(os.path.join(ROOT, "rala_amd", "host", "libassembly_graph.so"))
    L.io_paf_parse.restype = ctypes.c_void_p
    L.io_paf_parse.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int,
                               ctypes.c_uint32, ctypes.c_int]
    L.io_paf_size.restype = ctypes.c_uint64
    L.io_paf_size.argtypes = [ctypes.c_void_p]
    L.io_paf_ok.argtypes = [ctypes.c_void_p]
    L.io_paf_length_error.restype = ctypes.c_int64
    L.io_paf_length_error.argtypes = [ctypes.c_void_p]
    L.io_paf_copy.argtypes = [ctypes.c_void_p] * 9
    L.io_paf_free.argtypes = [ctypes.c_void_p]
    return L


def parse(path, names, read_len, threads, parallel, check_lengths=True):
    L = _lib()
    read_len = np.ascontiguousarray(read_len, dtype=np.uint32)
    h = L.io_paf_parse(path.encode(), "\n".join(names).encode(), read_len.ctypes.data, len(names), int(check_lengths),
                       threads, int(parallel))
    try:
        assert L.io_paf_ok(h)
        n = int(L.io_paf_size(h))
        cols = {f: np.zeros(n, dtype=np.uint32) for f in FIELDS}
        cols["strand"] = np.zeros(n, dtype=np.uint8)
        L.io_paf_copy(h, *[cols[f].ctypes.data for f in FIELDS], cols["strand"].ctypes.data)
        return cols, int(L.io_paf_length_error(h))
    finally:
        L.io_paf_free(h)


@pytest.fixture(params=["widest", "sse2"])
def tokenizer(request, monkeypatch):
    """both tokenizers of the parallel reader: 64-byte compares where the CPU has AVX-512BW, 16-byte otherwise"""
    if request.param == "sse2":
        monkeypatch.setenv("RALA_IO_NO_AVX512", "1")
    return request.param


@pytest.mark.parametrize("threads", [1, 2, 3, 8, 61])
def test_parallel_reader_matches_sequential(tmp_path, threads, tokenizer):
    ds = Dataset(3000, 600_000, 4)
    paf = str(tmp_path / "ovl.paf")
    ds.write_paf(paf)
    names = ["r%d" % i for i in range(ds.n_reads)]
    want, e0 = parse(paf, names, ds.read_len, 1, False)
    got, e1 = parse(paf, names, ds.read_len, threads, True)
    assert e0 == e1 == -1
    assert len(want["a_id"]) == len(ds.overlaps)
    for f in want:
        assert (want[f] == got[f]).all(), f
    # ... and both equal the generator's own columns
    for f in FIELDS:
        assert (got[f] == getattr(ds.overlaps, f)).all(), f
    assert (got["strand"] == ds.overlaps.strand).all()


def test_awkward_lines(tmp_path, tokenizer):
    lines = [
        "r0\t1000\t10\t900\t+\tr1\t2000\t5\t895\t800\t890\t255",
        "",                                                         # empty line
        "r1 comment\t2000\t0\t1500\t-\tr2\t3000\t100\t1600\t1400\t1500\t255\ttp:A:S\tcm:i:12",   # name cut at blank
        "rX\t500\t0\t400\t+\tr0\t1000\t0\t400\t300\t400\t60",       # unknown query
        "r2\t3000\t0\t100\t+\trY\t700\t0\t100\t90\t100\t0",         # unknown target
        "short\tline",                                              # fewer than 12 columns: skipped
        "r2\t3000\t7\t2000\t+\tr0\t1000\t1\t999\t900\t1993\t255\r", # CR LF
        "r0\t1000\t1\t2\t-\tr2\t3000\t3\t4\t5\t6\t7",               # no trailing newline
    ]
    path = str(tmp_path / "odd.paf")
    with open(path, "w") as f:
        f.write("\n".join(lines))
    names, lens = ["r0", "r1", "r2"], [1000, 2000, 3000]
    want, e0 = parse(path, names, lens, 1, False)
    for threads in (1, 2, 4):
        got, e1 = parse(path, names, lens, threads, True)
        assert e0 == e1 == -1
        for f in want:
            assert want[f].tolist() == got[f].tolist(), (threads, f)
    assert want["a_id"].tolist() == [0, 1, 0xFFFFFFFF, 2, 2, 0]
    assert want["b_id"].tolist() == [1, 2, 0, 0xFFFFFFFF, 0, 2]
    assert want["strand"].tolist() == [0, 1, 0, 0, 0, 1]
    assert want["length"].tolist() == [890, 1500, 400, 100, 1993, 6]


def test_lines_across_the_reader_s_buffers(tmp_path, tokenizer):
    """The reader streams the text through half-megabyte buffers: lines that straddle a buffer or a
    piece of the file, a line longer than a buffer, runs of empty lines at the seams."""
    rng = np.random.default_rng(3)
    names, lens = ["r%d" % i for i in range(50)], [1000 + i for i in range(50)]
    out = []
    for k in range(60_000):
        a, b = int(rng.integers(0, 50)), int(rng.integers(0, 50))
        line = "r%d\t%d\t%d\t%d\t%s\tr%d\t%d\t%d\t%d\t%d\t%d\t255" % (
            a, lens[a], k % 100, 500 + k % 400, "+-"[k & 1], b, lens[b], k % 90, 480 + k % 300, 400, 450 + k % 50)
        if k == 20_000:
            line += "\tzz:Z:" + "x" * 700_000                  # longer than a buffer
        if k % 7_000 == 0:
            line += "\tcg:Z:" + "5M" * int(rng.integers(1, 40_000))
        out.append(line)
        if rng.random() < 0.001:
            out.extend([""] * int(rng.integers(1, 5)))           # runs of empty lines
    path = str(tmp_path / "seams.paf")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n\n")
    assert os.path.getsize(path) > 3 << 20
    want, e0 = parse(path, names, lens, 1, False)
    assert len(want["a_id"]) == 60_000
    for threads in (1, 2, 3, 5, 8):
        got, e1 = parse(path, names, lens, threads, True)
        assert e0 == e1 == -1
        for f in want:
            assert (want[f] == got[f]).all(), (threads, f)


def test_length_mismatch_is_reported(tmp_path, tokenizer):
    path = str(tmp_path / "bad.paf")
    with open(path, "w") as f:
        f.write("r0\t1000\t0\t500\t+\tr1\t2000\t0\t500\t400\t500\t255\n" * 50)
        f.write("r1\t2000\t0\t500\t+\tr0\t1001\t0\t500\t400\t500\t255\n")     # target length wrong
        f.write("r0\t999\t0\t500\t+\tr1\t2000\t0\t500\t400\t500\t255\n")      # query length wrong
    for threads in (1, 3):
        _, seq = parse(path, ["r0", "r1"], [1000, 2000], 1, False)
        _, par = parse(path, ["r0", "r1"], [1000, 2000], threads, True)
        assert seq == par == 0                                   # the first offending line names read 0
        # the sensitive file goes through Overlap::transmute_, which checks no length at all
        # (overlap.cpp:84-114): neither bad line is an error there
        _, seq = parse(path, ["r0", "r1"], [1000, 2000], 1, False, check_lengths=False)
        _, par = parse(path, ["r0", "r1"], [1000, 2000], threads, True, check_lengths=False)
        assert seq == par == -1
