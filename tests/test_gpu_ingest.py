"""GPU: the device tokeniser (rala_hip_set_overlaps_from_paf, rala_amd/csrc/ingest_kernels.hip) gives the columns of the host
readers (rala_amd/host/io.cpp) for files that are lists of 12-column records - synthetic PAF, PAF shaped like minimap2's,
lines across the staging blocks and the chunks, tags of megabytes behind the columns - and says so when a file is something
else (the caller then takes the host reader); the CLI from PAF text to contigs with it and without it."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from rala_amd import build
from rala_amd.synth import Dataset

import test_ingest_cpu as host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = host.FIELDS


def _lib():
    build.build_host()
    L = ctypes.CDLL(os.path.join(ROOT, "rala_amd", "host", "librala_api.so"))
    L.hp_paf_device.restype = ctypes.c_void_p
    L.hp_paf_device.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32]
    L.hp_paf_device_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.hp_paf_device_copy.argtypes = [ctypes.c_void_p] * 9
    L.hp_paf_device_free.argtypes = [ctypes.c_void_p]
    return L


def device_parse(path, names, read_len, threads=4, check_lengths=True):
    """-> (columns or None, irregular flags, first read with a length mismatch)"""
    L = _lib()
    rl = np.ascontiguousarray(read_len, dtype=np.uint32)
    h = L.hp_paf_device(path.encode(), "\n".join(names).encode(), rl.ctypes.data, len(names), int(check_lengths), threads)
    try:
        info = np.zeros(6, dtype=np.int64)
        L.hp_paf_device_info(h, info.ctypes.data)
        assert info[0] == 0, info
        if info[1] or info[2] >= 0:
            return None, int(info[1]), int(info[2])
        n = int(info[3])
        cols = {f: np.zeros(n, dtype=np.uint32) for f in FIELDS}
        cols["strand"] = np.zeros(n, dtype=np.uint8)
        L.hp_paf_device_copy(h, *[cols[f].ctypes.data for f in FIELDS], cols["strand"].ctypes.data)
        return cols, 0, -1
    finally:
        L.hp_paf_device_free(h)


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("n,g,seed", [(3000, 600_000, 4), (20_000, 4_000_000, 8)])
def test_device_tokeniser_matches_host_reader(tmp_path, n, g, seed, threads):
    ds = Dataset(n, g, seed)
    paf = str(tmp_path / "ovl.paf")
    ds.write_paf(paf)
    names = ["r%d" % i for i in range(ds.n_reads)]
    got, irregular, bad = device_parse(paf, names, ds.read_len, threads)
    assert irregular == 0 and bad == -1
    for f in FIELDS:
        assert (got[f] == getattr(ds.overlaps, f)).all(), f
    assert (got["strand"] == ds.overlaps.strand).all()


def test_minimap2_shaped_paf_and_unknown_names(tmp_path):
    path, names, lens, want = host._minimap2_like(tmp_path)
    got, irregular, bad = device_parse(path, names, lens)
    assert irregular == 0 and bad == -1
    for f in want:
        assert got[f].tolist() == want[f], f
    # half of the names unknown to the table: their ids are RALA_HIP_NO_READ, the rest as the host reader gives them
    fewer = names[: len(names) // 2]
    want2, e = host.parse(path, fewer, lens[: len(fewer)], 2, True)
    got2, irregular, bad = device_parse(path, fewer, lens[: len(fewer)])
    assert e == -1 and irregular == 0 and bad == -1
    for f in want2:
        assert (got2[f] == want2[f]).all(), f
    assert (got2["a_id"] == 0xFFFFFFFF).any() and (got2["b_id"] == 0xFFFFFFFF).any()


def test_lines_across_blocks_and_long_tags(tmp_path):
    """tags of hundreds of kilobytes behind the twelve columns (a line that spans dozens of 16 KB chunks), runs of empty
    lines, names cut at a blank, CR LF, no newline at the end; a file of more than one 32 MB staging block"""
    rng = np.random.default_rng(3)
    names, lens = ["r%d" % i for i in range(50)], [1000 + i for i in range(50)]
    out = []
    for k in range(700_000):
        a, b = int(rng.integers(0, 50)), int(rng.integers(0, 50))
        line = "r%d%s\t%d\t%d\t%d\t%s\tr%d\t%d\t%d\t%d\t%d\t%d\t255" % (
            a, " some comment" if k % 1000 == 3 else "", lens[a], k % 100, 500 + k % 400, "+-"[k & 1], b, lens[b], k % 90, 480 + k % 300,
            400, 450 + k % 50)
        if k % 60_000 == 7:
            line += "\tzz:Z:" + "x" * 700_000
        if k % 9_000 == 0:
            line += "\tcg:Z:" + "5M" * int(rng.integers(1, 20_000))
        if k % 5_000 == 1:
            line += "\r"
        out.append(line)
        if rng.random() < 0.0005:
            out.extend([""] * int(rng.integers(1, 5)))
    path = str(tmp_path / "seams.paf")
    with open(path, "w") as f:
        f.write("\n".join(out))                                     # (no newline behind the last line)
    assert os.path.getsize(path) > 33 << 20
    want, e0 = host.parse(path, names, lens, 4, True)
    got, irregular, bad = device_parse(path, names, lens, threads=5)
    assert e0 == -1 and irregular == 0 and bad == -1
    assert len(got["a_id"]) == 700_000
    for f in want:
        assert (want[f] == got[f]).all(), f


def test_what_the_device_tokeniser_leaves_to_the_host_reader(tmp_path):
    good = "r0\t1000\t10\t900\t+\tr1\t2000\t5\t895\t800\t890\t255\n"
    names, lens = ["r0", "r1", "r2"], [1000, 2000, 3000]
    cases = {
        "short": good * 10 + "short\tline\n" + good,                       # fewer than 12 columns
        # the first eleven columns reach beyond the halo behind the chunk the line starts in (at byte 16 340 of 16 384)
        "name": good * 380 + "r0" + "x" * 3000 + good[2:] + good,
        "tiny": good + "a\n" * 5000 + good,                                  # more lines in a chunk than records can make
    }
    for what, text in cases.items():
        path = str(tmp_path / (what + ".paf"))
        open(path, "w").write(text)
        got, irregular, bad = device_parse(path, names, lens)
        assert got is None and irregular != 0, what
    # Overlap::transmute's length check: the first offending record in file order names its read
    path = str(tmp_path / "bad.paf")
    with open(path, "w") as f:
        f.write(good * 50)
        f.write("r1\t2000\t0\t500\t+\tr0\t1001\t0\t500\t400\t500\t255\n")     # target length wrong
        f.write("r0\t999\t0\t500\t+\tr1\t2000\t0\t500\t400\t500\t255\n")      # query length wrong
    got, irregular, bad = device_parse(path, names, lens)
    assert got is None and irregular == 0 and bad == 0
    got, irregular, bad = device_parse(path, names, lens, check_lengths=False)      # (Overlap::transmute_ checks nothing)
    assert irregular == 0 and bad == -1 and len(got["a_id"]) == 52
    # an empty file, a file of empty lines
    for what, text in (("empty", ""), ("blank", "\n\n\n")):
        path = str(tmp_path / (what + ".paf"))
        open(path, "w").write(text)
        got, irregular, bad = device_parse(path, names, lens)
        assert irregular == 0 and bad == -1 and len(got["a_id"]) == 0, what


def test_cli_with_and_without_the_device_tokeniser(tmp_path):
    """rala reads.fasta overlaps.paf: the same contigs whether the overlaps were tokenised on the device (the default for an
    uncompressed PAF on one GPU) or by the host reader (RALA_DEVICE_INGEST=0)"""
    build.build_host()
    exe = os.path.join(build.PKG, "host", "rala")
    ds = Dataset(3000, 400_000, 5)
    fa, paf = str(tmp_path / "reads.fasta"), str(tmp_path / "ovl.paf")
    ds.write_fasta(fa)
    ds.write_paf(paf)
    out = {}
    for mode in ("1", "0"):
        r = subprocess.run([exe, fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, RALA_DEVICE_INGEST=mode))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        out[mode] = (r.stdout, [x for x in r.stderr.decode().splitlines() if "number of" in x])
    assert out["1"][0] == out["0"][0] and len(out["1"][0]) > 1000
    assert out["1"][1] == out["0"][1]
