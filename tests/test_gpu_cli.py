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


def _read_fasta(path):
    names, seqs = [], []
    for line in open(path, "rb"):
        line = line.rstrip(b"\n")
        if line.startswith(b">"):
            names.append(line[1:].split()[0])
            seqs.append(b"")
        else:
            seqs[-1] += line
    return names, seqs


def _expected_layout(o, nodes, post, fa):
    """oracle graph after transitive reduction, with the reads' trimmed sequences, as an oracle
    layout object"""
    import layout

    names, seqs = _read_fasta(fa)
    p = o.piles()
    g = layout.oracle()
    for r in nodes[::2]:
        g.add_node_pair(int(r), names[r], seqs[r][int(p["begin"][r]): int(p["end"][r])])
    for s_, d_, l_ in zip(post["src"], post["dst"], post["len"]):
        g.add_edge(s_, d_, l_)
    for i in np.nonzero(post["marked"])[0]:
        if i % 2 == 0:
            g.mark_edge(i)
    g.note_transitive()
    g.remove_marked(False)
    return g


def _simplify(g):
    """Graph::simplify after the transitive reduction (reference graph.cpp:647-684); layout
    rounds with the seeds 0 .. 4 like rala::Graph.  Returns (tips, bubbles, long edges)."""
    count = {"tips": 0, "bubbles": 0, "long_edges": 0}

    def loop():
        while True:
            t, b = g.run("tips"), g.run("bubbles")
            count["tips"] += t
            count["bubbles"] += b
            if t + b == 0:
                break
    loop()
    g.run("shrink", 42)
    for seed in range(5):
        g.postprocess(seed)
        count["long_edges"] += g.run("long_edges")
        count["tips"] += g.run("tips")
    loop()
    return count["tips"], count["bubbles"], count["long_edges"]


# (2000, 1 Mb, seed 3): 20x coverage - tips, bubbles and long edges all occur (asserted below)
@pytest.mark.parametrize("n,genome,seed", [(600, 120_000, 17), (3000, 400_000, 5), (2000, 1_000_000, 3)])
def test_cli_layout_to_contigs(tmp_path, n, genome, seed):
    """rala <reads.fasta> <overlaps.paf>: construct on the GPU, simplify + unitigs on the host;
    the debug CSV and the contig FASTA against the oracle pipeline + oracle layout."""
    build.build_host()
    exe = os.path.join(build.PKG, "host", "rala")
    ds = Dataset(n, genome, seed)
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
    want = _expected_layout(o, nodes, post, fa)
    n_tips, n_bubbles, n_long = _simplify(want)

    prefix = str(tmp_path / "dbg")
    r = subprocess.run([exe, "-u", "-d", prefix, fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err = r.stderr.decode()
    assert r.returncode == 0, err
    assert "number of transitive edges = %d" % n_tr in err
    assert "number of tips = %d" % n_tips in err
    assert "number of bubbles = %d" % n_bubbles in err
    assert "number of long edges = %d" % n_long in err
    if genome == 1_000_000:
        assert n_tips > 0 and n_bubbles > 0 and n_long > 0, (n_tips, n_bubbles, n_long)
    assert "number of nodes = %d" % len(nodes) in err
    assert "number of edges = %d" % len(pre["src"]) in err
    # <prefix>.csv: the graph after simplify
    wn, we = want.dump()
    got = _edges_from_csv(prefix + ".csv")
    assert got == sorted((int(e), int(we["begin"][e]), int(we["end"][e]), int(we["length"][e]))
                         for e in np.nonzero(we["alive"])[0])
    # contigs: every forward node left after create_unitigs (-u keeps the short ones)
    want.run("unitigs")
    wn, _ = want.dump()
    exp = []
    for k in np.nonzero(wn["alive"])[0]:
        if k % 2 == 0:
            data = want.node_data(int(k))
            exp.append((b">Ctg%d RC:i:%d LN:i:%d" % (len(exp), wn["n_seq"][k], len(data)), data))
    lines = r.stdout.split(b"\n")
    got_c = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
    assert got_c == exp
    assert max(len(d) for _, d in exp) > (0.5 if genome < 1_000_000 else 0.05) * genome     # most of the genome in one contig (20x: several)
    # without -u only contigs of >= 6 reads and >= 10 kb remain (graph.cpp:2053-2054)
    r2 = subprocess.run([exe, fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r2.returncode == 0
    kept = [(h, d) for h, d in exp if int(h.split(b"RC:i:")[1].split()[0]) >= 6 and len(d) >= 10000]
    lines = r2.stdout.split(b"\n")
    got_k = [lines[i + 1] for i in range(0, len(lines) - 1, 2)]
    assert got_k == [d for _, d in kept]


def test_cli_preconstruct(tmp_path):
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
    nodes = o.nodes()
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


def test_cli_other_formats(tmp_path):
    """MHAP overlaps (reference overlap.cpp:12-20: 1-based ids, strand = a_rc != b_rc, length = the
    longer span), gzip-compressed inputs and FASTQ reads give the same contigs as FASTA + PAF."""
    import gzip

    build.build_host()
    exe = os.path.join(build.PKG, "host", "rala")
    ds = Dataset(600, 120_000, 17)
    fa = str(tmp_path / "reads.fasta")
    paf = str(tmp_path / "ovl.paf")
    ds.write_fasta(fa)
    ds.write_paf(paf)
    want = subprocess.run([exe, "-u", fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert want.returncode == 0 and want.stdout.count(b">Ctg") > 0

    # PAF -> MHAP: "a_id b_id error minmers a_rc a_begin a_end a_len b_rc b_begin b_end b_len", ids from 1
    mhap = str(tmp_path / "ovl.mhap")
    with open(paf) as src, open(mhap, "w") as dst:
        for line in src:
            f = line.split("\t")
            a, b = int(f[0][1:]) + 1, int(f[5][1:]) + 1
            rc = 0 if f[4] == "+" else 1
            assert int(f[10]) == max(int(f[3]) - int(f[2]), int(f[8]) - int(f[7]))       # PAF col 11 = longer span
            dst.write("%d %d 0.1 42 0 %s %s %s %d %s %s %s\n" % (a, b, f[2], f[3], f[1], rc, f[7], f[8], f[6]))
    got = subprocess.run([exe, "-u", fa, mhap], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr.decode()
    assert got.stdout == want.stdout

    # gzip: reads and overlaps
    for path in (fa, paf, mhap):
        with open(path, "rb") as src, gzip.open(path + ".gz", "wb") as dst:
            dst.write(src.read())
    got = subprocess.run([exe, "-u", fa + ".gz", paf + ".gz"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr.decode()
    assert got.stdout == want.stdout
    got = subprocess.run([exe, "-u", fa + ".gz", mhap + ".gz"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0 and got.stdout == want.stdout

    # FASTQ reads (four-line records)
    fq = str(tmp_path / "reads.fastq")
    names, seqs = _read_fasta(fa)
    with open(fq, "wb") as dst:
        for nm, sq in zip(names, seqs):
            dst.write(b"@" + nm + b"\n" + sq + b"\n+\n" + b"I" * len(sq) + b"\n")
    got = subprocess.run([exe, "-u", fq, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr.decode()
    assert got.stdout == want.stdout

    # a PAF whose lengths disagree with the reads is fatal, with the reference's message
    bad = str(tmp_path / "bad.paf")
    with open(paf) as src, open(bad, "w") as dst:
        lines = src.readlines()
        f = lines[5].split("\t")
        f[1] = str(int(f[1]) + 1)
        lines[5] = "\t".join(f)
        dst.writelines(lines)
    got = subprocess.run([exe, fa, bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 1 and b"unequal lengths in sequence and overlap file" in got.stderr


def test_cli_sensitive_pass(tmp_path):
    """rala -s <sensitive overlaps>: the two-pass workflow of the reference (main.cpp:76-94) - the
    second overlap set is relative to the trimmed reads of the first pass."""
    build.build_host()
    exe = os.path.join(build.PKG, "host", "rala")
    n, genome, seed = 6000, 1_600_000, 19
    ds = Dataset(n, genome, seed)
    fa = str(tmp_path / "reads.fasta")
    paf = str(tmp_path / "ovl.paf")
    sens_paf = str(tmp_path / "sens.paf")
    ds.write_fasta(fa)
    ds.write_paf(paf)

    o = Oracle(ds.read_len, ds.overlaps, n_threads=8)
    assert o.initialize() == 0
    o.pass2()
    o.preprocess_chimeras()
    p = o.piles()
    sens = ds.sensitive(p["alive"], p["begin"], p["end"])
    assert len(sens) > 0
    ds.write_paf(sens_paf, sensitive=True, target_len=(p["end"] - p["begin"]).astype(np.uint32))
    o.preprocess_repeats(sens)
    assert len(o.all_intervals(2)[1]) > 0             # the data set has repeat hills
    o.build_graph()
    pre = o.edges()
    n_tr = o.remove_transitive_edges()
    post = o.edges()
    nodes = o.nodes()
    want = _expected_layout(o, nodes, post, fa)
    _simplify(want)
    want.run("unitigs")
    wn, _ = want.dump()
    exp = [want.node_data(int(k)) for k in np.nonzero(wn["alive"])[0] if k % 2 == 0]

    r = subprocess.run([exe, "-u", "-s", sens_paf, fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err = r.stderr.decode()
    assert r.returncode == 0, err
    assert "number of transitive edges = %d" % n_tr in err
    assert "number of edges = %d" % len(pre["src"]) in err
    lines = r.stdout.split(b"\n")
    assert [lines[i + 1] for i in range(0, len(lines) - 1, 2)] == exp


def test_reference_cli_drop_in(tmp_path):
    """the reference's own src/main.cpp, compiled unchanged from /root/reference
    against rala_amd/host's headers (rala_amd/build.py: build_reference_cli; the binary travels to
    the GPU box), gives the same contigs, debug CSV and -p reads as this package's own driver."""
    build.build_host()
    ref = os.path.join(build.PKG, "host", "_refcli", "rala_ref")
    if not os.path.exists(ref):
        pytest.skip("reference CLI was not built (no /root/reference where the build ran)")
    exe = os.path.join(build.PKG, "host", "rala")
    ds = Dataset(1200, 200_000, 23)
    fa, paf = str(tmp_path / "reads.fasta"), str(tmp_path / "ovl.paf")
    ds.write_fasta(fa)
    ds.write_paf(paf)
    outs = {}
    for name, binary in (("ours", exe), ("ref", ref)):
        prefix = str(tmp_path / ("dbg_" + name))
        r = subprocess.run([binary, "-u", "-d", prefix, "-t", "4", fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        p = subprocess.run([binary, "-p", fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        outs[name] = (r.stdout, open(prefix + ".csv").read(), open(prefix + ".json").read(), p.stdout)
    assert outs["ours"][0] == outs["ref"][0] and len(outs["ours"][0]) > 100_000
    assert outs["ours"][1] == outs["ref"][1]
    assert outs["ours"][2] == outs["ref"][2]
    assert outs["ours"][3] == outs["ref"][3] and outs["ours"][3].count(b">") > 10


@pytest.mark.parametrize("gpus", [2, 3])
def test_cli_several_ranks_give_the_single_gpu_result(tmp_path, gpus):
    """rala --gpus N (rala::Graph over N rank objects, rala_hip_mg_*): contigs, debug CSV / JSON and
    the -p reads equal the one-GPU run's, with and without -s.  On this one-GPU box the ranks
    share device 0 and talk through the in-process transport (RALA_COMM=local)."""
    build.build_host()
    exe = os.path.join(build.PKG, "host", "rala")
    n, genome, seed = 6000, 1_600_000, 19
    ds = Dataset(n, genome, seed)
    fa, paf, sens_paf = str(tmp_path / "reads.fasta"), str(tmp_path / "ovl.paf"), str(tmp_path / "sens.paf")
    ds.write_fasta(fa)
    ds.write_paf(paf)
    o = Oracle(ds.read_len, ds.overlaps, n_threads=8)
    assert o.initialize() == 0
    o.pass2()
    o.preprocess_chimeras()
    p = o.piles()
    sens = ds.sensitive(p["alive"], p["begin"], p["end"])
    ds.write_paf(sens_paf, sensitive=True, target_len=(p["end"] - p["begin"]).astype(np.uint32))
    many = dict(os.environ, RALA_COMM="local", RALA_GPU_DEVICES=",".join(["0"] * gpus))

    def run(args, env, tag):
        prefix = str(tmp_path / tag)
        r = subprocess.run([exe, "-u", "-d", prefix] + args + [fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()
        return r.stdout, open(prefix + ".csv").read(), open(prefix + ".json").read()

    # (round 5) every rank tokenises its own byte range of the PAF text - and its share of the -s file - on its own GPU;
    # RALA_DEVICE_INGEST=0: the host readers hand out slices from host memory, as before
    host_readers = dict(many, RALA_DEVICE_INGEST="0")
    for extra in ([], ["-s", sens_paf]):
        one = run(extra, os.environ, "one%d" % len(extra))
        for env, tag in ((many, "many"), (host_readers, "host")):
            more = run(extra + ["--gpus", str(gpus)], env, "%s%d" % (tag, len(extra)))
            assert one[0] == more[0] and len(one[0]) > 100_000
            assert one[1] == more[1]
            assert one[2] == more[2]            # the JSON carries whole coverage vectors: fetched from the owners
        if extra:
            old = run(extra, dict(os.environ, RALA_DEVICE_INGEST="0"), "onehost")   # one GPU, the -s file through the host reader
            assert old == one
    a = subprocess.run([exe, "-p", fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([exe, "-p", "--gpus", str(gpus), fa, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=many)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout


@pytest.mark.skipif(os.environ.get("RALA_SKIP_C5") == "1", reason="RALA_SKIP_C5=1")
def test_cli_c5x_eight_ranks_sensitive_to_contigs(tmp_path):
    """BASELINE configs[4]'s shape at the size an oracle can hold: c5x (200 k reads at 75x, 15 M overlaps) through
    `rala -s <sensitive.paf> --gpus 8` - sharded construct with the sensitive pass, transitive reduction, simplify
    with the layout kernel, unitigs, contig FASTA - against the oracle pipeline (reference Pile / Overlap objects
    where built) + the oracle layout.  The eight ranks share this box's one GPU (in-process transport)."""
    import time

    build.build_host()
    exe = os.path.join(build.PKG, "host", "rala")
    ds = Dataset.config("c5x")
    fa, paf, sens_paf = str(tmp_path / "reads.fasta"), str(tmp_path / "ovl.paf"), str(tmp_path / "sens.paf")
    t0 = time.time()
    ds.write_fasta(fa)
    ds.write_paf(paf)
    from rala_amd.cpus import effective_cpus
    o = Oracle(ds.read_len, ds.overlaps, n_threads=effective_cpus())
    assert o.initialize() == 0
    o.pass2()
    o.preprocess_chimeras()
    p = o.piles()
    sens = ds.sensitive(p["alive"], p["begin"], p["end"])
    assert len(sens) > 100_000
    ds.write_paf(sens_paf, sensitive=True, target_len=(p["end"] - p["begin"]).astype(np.uint32))
    o.preprocess_repeats(sens)
    o.build_graph()
    pre = o.edges()
    n_tr = o.remove_transitive_edges()
    post = o.edges()
    nodes = o.nodes()
    want = _expected_layout(o, nodes, post, fa)
    n_tips, n_bubbles, n_long = _simplify(want)
    t1 = time.time()

    env = dict(os.environ, RALA_COMM="local", RALA_GPU_DEVICES=",".join(["0"] * 8))
    prefix = str(tmp_path / "dbg")
    r = subprocess.run([exe, "-u", "-d", prefix, "-s", sens_paf, "--gpus", "8", "-t", str(effective_cpus()), fa, paf],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    err = r.stderr.decode()
    assert r.returncode == 0, err[-3000:]
    print("c5x: oracle %.0f s, rala --gpus 8 %.0f s; %d nodes, %d edges, %d transitive, %d tips, %d bubbles, %d long edges" % (
        t1 - t0, time.time() - t1, len(nodes), len(pre["src"]), n_tr, n_tips, n_bubbles, n_long))
    assert "number of transitive edges = %d" % n_tr in err
    assert "number of tips = %d" % n_tips in err
    assert "number of bubbles = %d" % n_bubbles in err
    assert "number of long edges = %d" % n_long in err
    assert "number of nodes = %d" % len(nodes) in err
    assert "number of edges = %d" % len(pre["src"]) in err
    wn, we = want.dump()
    got = _edges_from_csv(prefix + ".csv")
    assert got == sorted((int(e), int(we["begin"][e]), int(we["end"][e]), int(we["length"][e]))
                         for e in np.nonzero(we["alive"])[0])
    want.run("unitigs")
    wn, _ = want.dump()
    exp = []
    for k in np.nonzero(wn["alive"])[0]:
        if k % 2 == 0:
            data = want.node_data(int(k))
            exp.append((b">Ctg%d RC:i:%d LN:i:%d" % (len(exp), wn["n_seq"][k], len(data)), data))
    lines = r.stdout.split(b"\n")
    got_c = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
    assert got_c == exp
    assert sum(len(d) for _, d in exp) > 10_000_000        # tens of megabases of contigs out of a 26.65 Mb genome
