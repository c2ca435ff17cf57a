"""GPU: the rala command line (rala_amd/host, the reference's Graph / Pile / Overlap /
Sequence interface over librala_hip) on FASTA + PAF files."""
import os
import subprocess

import numpy as np
import pytest

from rala_amd import build
from rala_amd.synth import Dataset
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _edges_from_csv(path):
    out = []
    for line in open(path):
        f = line.rstrip("\n").split(",")
        if f[2] != "1":
            continue
        a = int(f[0].split()[0]); b = int(f[1].split()[0])
        eid, length, _w = f[3].split()
        out.append((int(eid), a, b, int(length)))
    return sorted(out)


def test_cli_debug_csv_and_preconstruct(tmp_path):
    build.build_host()
    exe = os.path.join(build.PKG, "host", "rala")
    ds = Dataset(600, 120_000, 17)
    fa = str(tmp_path / "reads.fasta")
    paf = str(tmp_path / "ovl.paf")
    ds.write_fasta(fa)
    ds.write_paf(paf)

    o = Oracle(ds.read_len, ds.overlaps, n_threads=4)
    assert o.construct() == 0
    pre = o.edges()
    n_tr = o.remove_transitive_edges()
    post = o.edges()
    nodes = o.nodes()

    # full run with debug output: <prefix>.csv holds the graph after transitive reduction
    prefix = str(tmp_path / "dbg")
    r = subprocess.run([exe, "-u", "-d", prefix, fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert "number of transitive edges = %d" % n_tr in r.stderr
    assert "number of nodes = %d" % len(nodes) in r.stderr
    assert "number of edges = %d" % len(pre["src"]) in r.stderr
    got = _edges_from_csv(prefix + ".csv")
    keep = post["marked"] == 0
    want = sorted((int(e), int(post["src"][e]), int(post["dst"][e]), int(post["len"][e]))
                  for e in np.nonzero(keep)[0])
    assert got == want
    # -u prints every forward node as a contig record
    assert r.stdout.count(">Ctg") == len(nodes) // 2

    # -p: uncontained reads that still have edges, trimmed to their valid regions
    r = subprocess.run([exe, "-p", fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    names = [l[1:] for l in r.stdout.splitlines() if l.startswith(">")]
    seqs = [l for l in r.stdout.splitlines() if not l.startswith(">")]
    p = o.piles()
    with_edges = set()
    for e in range(len(pre["src"])):
        with_edges.add(int(nodes[pre["src"][e]])); with_edges.add(int(nodes[pre["dst"][e]]))
    assert sorted(names) == sorted("r%d" % r_ for r_ in with_edges)
    for nm, sq in zip(names, seqs):
        rid = int(nm[1:])
        assert len(sq) == int(p["end"][rid]) - int(p["begin"][rid])


def test_cli_rejects_unknown_extension(tmp_path):
    build.build_host()
    exe = os.path.join(build.PKG, "host", "rala")
    r = subprocess.run([exe, "reads.txt", "ovl.paf"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "unsupported format extension" in r.stderr
