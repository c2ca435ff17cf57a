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


def _to_mhap(paf, mhap):
    """PAF -> MHAP: "a_id b_id error minmers a_rc a_begin a_end a_len b_rc b_begin b_end b_len", ids from 1"""
    with open(paf) as src, open(mhap, "w") as dst:
        for line in src:
            f = line.split("\t")
            if len(f) < 12:
                dst.write(line)
                continue
            a, b = int(f[0][1:]) + 1, int(f[5][1:]) + 1
            dst.write("%d %d 0.1 42 0 %s %s %s %d %s %s %s\n" % (a, b, f[2], f[3], f[1], 0 if f[4] == "+" else 1, f[7], f[8], f[6]))


@pytest.mark.parametrize("threads", [1, 2, 4, 9])
def test_streamed_reader_gzip_and_mhap(tmp_path, threads, tokenizer):
    """compressed PAF and MHAP go through one inflating thread + parser threads (io::read_overlaps_streamed):
    the same columns as the line-by-line readers, whatever the thread count; blocks of 4 MB are cut at lines"""
    import gzip
    import shutil

    ds = Dataset(6000, 1_000_000, 4)            # 18 MB of PAF: several blocks
    paf = str(tmp_path / "ovl.paf")
    ds.write_paf(paf)
    assert os.path.getsize(paf) > 9 << 20
    names = ["r%d" % i for i in range(ds.n_reads)]
    want, e0 = parse(paf, names, ds.read_len, 1, 0)
    gz = paf + ".gz"
    with open(paf, "rb") as src, gzip.open(gz, "wb", compresslevel=1) as dst:
        shutil.copyfileobj(src, dst)
    for path in (paf, gz):
        got, e1 = parse(path, names, ds.read_len, threads, 2)
        assert e0 == e1 == -1
        for f in want:
            assert (want[f] == got[f]).all(), (path, f)
    mhap = str(tmp_path / "ovl.mhap")
    _to_mhap(paf, mhap)
    with open(mhap, "rb") as src, gzip.open(mhap + ".gz", "wb", compresslevel=1) as dst:
        shutil.copyfileobj(src, dst)
    seq, e2 = parse(mhap, names, ds.read_len, 1, 4)
    for path in (mhap, mhap + ".gz"):
        got, e3 = parse(path, names, ds.read_len, threads, 3)
        assert e2 == e3 == -1
        for f in seq:
            assert (seq[f] == got[f]).all(), (path, f)
    # MHAP carries no alignment length: the longer span (overlap.cpp:18)
    for f in ("a_id", "b_id", "a_begin", "a_end", "b_begin", "b_end"):
        assert (seq[f] == want[f]).all(), f
    assert (seq["strand"] == want["strand"]).all()
    assert (seq["length"] == np.maximum(want["a_end"] - want["a_begin"], want["b_end"] - want["b_begin"])).all()


def test_streamed_reader_awkward_input(tmp_path, tokenizer):
    """empty lines, short lines, CR LF, a missing last newline, a line longer than a block, length mismatches"""
    import gzip

    names, lens = ["r0", "r1", "r2"], [1000, 2000, 3000]
    lines = [
        "r0\t1000\t10\t900\t+\tr1\t2000\t5\t895\t800\t890\t255",
        "",
        "r1 comment\t2000\t0\t1500\t-\tr2\t3000\t100\t1600\t1400\t1500\t255\ttp:A:S\tzz:Z:" + "y" * (5 << 20),
        "rX\t500\t0\t400\t+\tr0\t1000\t0\t400\t300\t400\t60",
        "short\tline",
        "r2\t3000\t7\t2000\t+\tr0\t1000\t1\t999\t900\t1993\t255\r",
        "r0\t1000\t1\t2\t-\tr2\t3000\t3\t4\t5\t6\t7",
    ]
    path = str(tmp_path / "odd.paf.gz")
    with gzip.open(path, "wt") as f:
        f.write("\n".join(lines))
    want, e0 = parse(path, names, lens, 1, 0)
    for threads in (1, 3):
        got, e1 = parse(path, names, lens, threads, 2)
        assert e0 == e1 == -1
        for f in want:
            assert want[f].tolist() == got[f].tolist(), (threads, f)
    assert want["a_id"].tolist() == [0, 1, 0xFFFFFFFF, 2, 0]
    bad = str(tmp_path / "bad.mhap")
    with open(bad, "w") as f:
        f.write("1 2 0.1 42 0 0 500 1000 0 0 500 2000\n" * 30)
        f.write("2 1 0.1 42 0 0 500 2000 1 0 500 1001\n")     # target length wrong: read 0
        f.write("1 2 0.1 42 0 0 500 999 0 0 500 2000\n")
    for mode, threads in ((4, 1), (3, 1), (3, 4)):
        _, e = parse(bad, names, lens, threads, mode)
        assert e == 0
        _, e = parse(bad, names, lens, threads, mode, check_lengths=False)
        assert e == -1


def _minimap2_like(tmp_path, n_lines=4000, seed=11):
    """A PAF shaped like `minimap2 -x ava-ont reads.fq reads.fq` output: read names as sequencers write them, the
    twelve mandatory columns, then minimap2's tags in its order (tp cm s1 [s2] dv rl, sometimes cg); self hits,
    both orders of a pair, repeated pairs, secondary hits, mapping quality 0.  Returns (path, names, lengths, want)
    with `want` parsed here in Python (tab split, name = the query / target column as it stands)."""
    rng = np.random.default_rng(seed)
    n = 60
    names = []
    for i in range(n):
        kind = i % 3
        if kind == 0:      # nanopore: a UUID
            h = "%032x" % int(rng.integers(0, 2 ** 62))
            names.append("%s-%s-%s-%s-%s" % (h[:8], h[8:12], h[12:16], h[16:20], h[20:32]))
        elif kind == 1:    # pacbio: movie/zmw/ccs
            names.append("m54238_180901_011437/%d/ccs" % int(rng.integers(4_000_000, 5_000_000)))
        else:              # illumina-ish / SRA
            names.append("SRR%d.%d" % (7_000_000 + i, int(rng.integers(1, 10 ** 6))))
    lens = [int(x) for x in rng.integers(3000, 40000, n)]
    want = {f: [] for f in FIELDS + ("strand",)}
    lines = []
    for k in range(n_lines):
        a = int(rng.integers(0, n))
        b = a if k % 97 == 0 else int(rng.integers(0, n))          # self hits
        span = int(rng.integers(500, min(lens[a], lens[b])))
        qb = int(rng.integers(0, lens[a] - span + 1)); qe = qb + span
        tb = int(rng.integers(0, lens[b] - span + 1)); te = tb + span - int(rng.integers(0, 20))
        strand = "+-"[int(rng.integers(0, 2))]
        alen = max(qe - qb, te - tb) + int(rng.integers(0, 50))
        match = int(alen * 0.85)
        mapq = 0 if k % 13 == 0 else int(rng.integers(1, 61))
        tags = ["tp:A:%s" % ("S" if k % 11 == 0 else "P"), "cm:i:%d" % int(rng.integers(5, 900)),
                "s1:i:%d" % int(rng.integers(50, 9000))]
        if k % 5:
            tags.append("s2:i:%d" % int(rng.integers(0, 5000)))
        tags += ["dv:f:%.4f" % float(rng.random() * 0.2), "rl:i:%d" % int(rng.integers(0, 3000))]
        if k % 17 == 0:
            tags.append("cg:Z:" + "".join("%d%s" % (int(rng.integers(1, 400)), "MID"[int(rng.integers(0, 3))])
                                          for _ in range(int(rng.integers(1, 300)))))
        lines.append("\t".join([names[a], str(lens[a]), str(qb), str(qe), strand, names[b], str(lens[b]), str(tb), str(te),
                                str(match), str(alen), str(mapq)] + tags))
        if k % 401 == 0:
            lines.append(lines[-1])                                  # the same hit twice
        reps = 2 if k % 401 == 0 else 1
        for _ in range(reps):
            for f, v in zip(FIELDS, (a, b, qb, qe, tb, te, alen)):
                want[f].append(v)
            want["strand"].append(1 if strand == "-" else 0)
    path = str(tmp_path / "ava.paf")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path, names, lens, want


@pytest.mark.parametrize("threads", [1, 5])
def test_minimap2_shaped_paf(tmp_path, threads, tokenizer):
    """every reader (line by line, parallel, streamed plain and gzip) against a parse done here in Python"""
    import gzip
    import shutil

    path, names, lens, want = _minimap2_like(tmp_path)
    with open(path, "rb") as src, gzip.open(path + ".gz", "wb", compresslevel=6) as dst:
        shutil.copyfileobj(src, dst)
    for p, mode in ((path, 0), (path, 1), (path, 2), (path + ".gz", 0), (path + ".gz", 2)):
        got, e = parse(p, names, lens, threads, mode)
        assert e == -1, (p, mode)
        for f in want:
            assert got[f].tolist() == want[f], (p, mode, f)


def _write_bgzf(src_path, dst_path, block=60000, level=1):
    """what `bgzip` writes: gzip members of at most 64 KB with their compressed size in a 'BC' extra field, then the
    empty end-of-file block (SAM specification, section 4.1)"""
    import struct
    import zlib

    def member(data):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(data) + c.flush()
        total = 18 + len(body) + 8
        return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, total - 1) +
                body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    with open(src_path, "rb") as src, open(dst_path, "wb") as dst:
        while True:
            data = src.read(block)
            if not data:
                break
            dst.write(member(data))
        dst.write(member(b""))


@pytest.mark.parametrize("threads", [1, 2, 7, 16])
def test_bgzf_is_inflated_by_several_threads(tmp_path, threads, tokenizer):
    """a bgzip'ed PAF / MHAP (blocks inflated in parallel, given out in file order) = the plain file; a block whose
    checksum is wrong and a file that ends inside a block are errors"""
    import gzip

    ds = Dataset(3000, 600_000, 6)
    names = ["r%d" % i for i in range(ds.n_reads)]
    paf = str(tmp_path / "ovl.paf")
    ds.write_paf(paf)
    want, e0 = parse(paf, names, ds.read_len, 1, 0)
    bg = paf + ".gz"
    _write_bgzf(paf, bg, block=int(np.random.default_rng(threads).integers(500, 65000)))
    with gzip.open(bg, "rb") as f:                       # (a valid multi-member gzip file for anybody else)
        assert len(f.read()) == os.path.getsize(paf)
    got, e1 = parse(bg, names, ds.read_len, threads, 2)
    assert e0 == e1 == -1
    for f in want:
        assert (want[f] == got[f]).all(), f
    mhap = str(tmp_path / "ovl.mhap")
    _to_mhap(paf, mhap)
    _write_bgzf(mhap, mhap + ".gz")
    seq, _ = parse(mhap, names, ds.read_len, 1, 4)
    got, _ = parse(mhap + ".gz", names, ds.read_len, threads, 3)
    for f in seq:
        assert (seq[f] == got[f]).all(), f
    # a flipped byte in the middle of a block: the checksum (or the deflate stream) fails
    raw = bytearray(open(bg, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = str(tmp_path / "bad.paf.gz")
    open(bad, "wb").write(bytes(raw))
    L = _lib()
    rl = np.ascontiguousarray(ds.read_len, dtype=np.uint32)
    h = L.io_paf_parse(bad.encode(), "\n".join(names).encode(), rl.ctypes.data, len(names), 1, threads, 2)
    assert not L.io_paf_ok(h)
    L.io_paf_free(h)
    cut = str(tmp_path / "cut.paf.gz")
    open(cut, "wb").write(bytes(open(bg, "rb").read()[:os.path.getsize(bg) * 2 // 3]))
    h = L.io_paf_parse(cut.encode(), "\n".join(names).encode(), rl.ctypes.data, len(names), 1, threads, 2)
    assert not L.io_paf_ok(h)
    L.io_paf_free(h)


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("mode", [2, 3])
def test_cut_gzip_stream_is_an_error(tmp_path, threads, mode):
    """a plain (single-member) .paf.gz / .mhap.gz that ends inside its deflate stream - a download cut short - is a
    broken file, as a cut BGZF file is; gzread reports it as a short read, only the stream's state says why
    (half a file used to parse as 15 384 of 31 117 records)"""
    import gzip

    ds = Dataset(3000, 600_000, 6)
    names = ["r%d" % i for i in range(ds.n_reads)]
    paf = str(tmp_path / "ovl.paf")
    ds.write_paf(paf)
    src = paf
    if mode == 3:
        src = str(tmp_path / "ovl.mhap")
        _to_mhap(paf, src)
    gz = src + ".gz"
    with open(src, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as dst:
        dst.write(f.read())
    whole, e = parse(gz, names, ds.read_len, threads, mode)
    assert e == -1 and len(whole["a_id"]) > 10_000
    L = _lib()
    rl = np.ascontiguousarray(ds.read_len, dtype=np.uint32)
    size = os.path.getsize(gz)
    for keep in (size // 2, size - 9, size - 1):              # inside the stream; inside / one byte short of the trailer
        cut = str(tmp_path / ("cut%d%s" % (keep, ".paf.gz" if mode == 2 else ".mhap.gz")))
        open(cut, "wb").write(open(gz, "rb").read()[:keep])
        h = L.io_paf_parse(cut.encode(), "\n".join(names).encode(), rl.ctypes.data, len(names), 1, threads, mode)
        assert not L.io_paf_ok(h), keep
        L.io_paf_free(h)
