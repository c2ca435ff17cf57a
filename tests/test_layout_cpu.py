"""CPU: the clean-up stages after transitive reduction (tips, bubbles, unitigs, shrink; SURVEY.md
section 8(f) rank 1) - product (index based) against the oracle restatement (pointer based, the
reference's object model), on graphs the oracle pipeline builds from synthetic data and on
hand-made shapes."""
import numpy as np
import pytest

from rala_amd.synth import Dataset

import layout
import parity


def _random_dna(rng, n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))


def _both():
    return layout.product(), layout.oracle()


def _from_pipeline(n, g, seed):
    """graph after construct + transitive reduction from the oracle pipeline, loaded into both"""
    ds = Dataset(n, g, seed)
    st = parity.oracle_stages(ds)
    piles = st["piles2"]
    rng = np.random.default_rng(seed)
    graphs = _both()
    for k, r in enumerate(st["nodes"][::2]):
        data = _random_dna(rng, int(piles["end"][r] - piles["begin"][r]))
        for gph in graphs:
            gph.add_node_pair(int(r), b"read%d" % r, data)
    e = st["edges"]
    for gph in graphs:
        for s, d, l in zip(e["src"], e["dst"], e["len"]):
            gph.add_edge(s, d, l)
        for i in np.nonzero(e["marked"])[0]:
            if i % 2 == 0:                       # marks come in twin pairs
                gph.mark_edge(i)
        gph.note_transitive()
        gph.remove_marked(False)
    return graphs


def _simplify(gph, log, engine=None):
    """Graph::simplify after the transitive reduction (reference graph.cpp:647-684); the layout
    rounds take the seeds 0 .. 4 like rala::Graph does"""
    def loop():
        while True:
            t = gph.run("tips")
            b = gph.run("bubbles")
            log.append(("loop", t, b))
            if t + b == 0:
                break
    loop()
    log.append(("shrink", gph.run("shrink", 42)))
    for seed in range(5):
        gph.postprocess(seed, engine)
        log.append(("round", gph.run("long_edges"), gph.run("tips")))
    loop()
    log.append(("unitigs", gph.run("unitigs")))


@pytest.mark.parametrize("n,g,seed", [(3000, 600_000, 21), (5000, 1_000_000, 7), (6000, 1_600_000, 19),
                                      (4000, 400_000, 2)])
def test_simplify_matches_oracle(n, g, seed):
    prod, ora = _from_pipeline(n, g, seed)
    layout.assert_same_graph(prod, ora, "after transitive reduction")
    la, lb = [], []
    _simplify(prod, la)
    _simplify(ora, lb)
    assert la == lb
    layout.assert_same_graph(prod, ora, "after simplify")
    nodes, _ = prod.dump()
    assert nodes["alive"].sum() > 0
    assert any(x[0] == "unitigs" and x[1] > 0 for x in la)


def test_simplify_with_tips_bubbles_and_long_edges():
    """a 20x data set on which every clean-up stage has work to do (most data sets of the suite
    are so well covered that tips, bubbles and long edges do not occur): counts equal and > 0"""
    prod, ora = _from_pipeline(2000, 1_000_000, 3)
    la, lb = [], []
    _simplify(prod, la)
    _simplify(ora, lb)
    assert la == lb
    layout.assert_same_graph(prod, ora, "after simplify")
    tips = sum(x[1] for x in la if x[0] == "loop") + sum(x[2] for x in la if x[0] == "round")
    bubbles = sum(x[2] for x in la if x[0] == "loop")
    long_edges = sum(x[1] for x in la if x[0] == "round")
    assert tips > 0 and bubbles > 0 and long_edges > 0, (tips, bubbles, long_edges)


@pytest.mark.parametrize("op,arg", [("tips", 0), ("bubbles", 0), ("unitigs", 0), ("shrink", 3), ("shrink", 42)])
@pytest.mark.parametrize("n,g,seed", [(3000, 600_000, 21), (2000, 1_200_000, 33)])
def test_single_stage_matches_oracle(n, g, seed, op, arg):
    prod, ora = _from_pipeline(n, g, seed)
    assert prod.run(op, arg) == ora.run(op, arg)
    layout.assert_same_graph(prod, ora, op)


def _chain(graphs, rng, n_reads, step=3000, length=10000, first_seq=0):
    """a linear chain of reads, each overlapping the next: nodes 2k (forward), 2k + 1 (reverse)"""
    base = None
    for gph in graphs:
        nodes0 = gph.dump()[0]["alive"].shape[0]
        edges0 = gph.dump()[1]["alive"].shape[0]
        r = np.random.default_rng(1234 + first_seq)
        for k in range(n_reads):
            gph.add_node_pair(first_seq + k, b"r%d" % (first_seq + k), _random_dna(r, length))
        for k in range(n_reads - 1):
            a, b = nodes0 + 2 * k, nodes0 + 2 * (k + 1)
            gph.add_edge(a, b, step)               # forward strand
            gph.add_edge(b + 1, a + 1, step)       # its twin on the reverse strand
        base = (nodes0, edges0)
    return base


def test_hand_made_shapes():
    rng = np.random.default_rng(0)
    # 1. plain chain: one unitig pair, 3000 * 11 + 10000 bases
    graphs = _both()
    _chain(graphs, rng, 12)
    for gph in graphs:
        assert gph.run("unitigs") == 1
    layout.assert_same_graph(*graphs, "chain")
    nodes, edges = graphs[0].dump()
    assert nodes["alive"].sum() == 2 and nodes["length"][nodes["alive"] == 1].tolist() == [43000, 43000]
    assert nodes["n_seq"][nodes["alive"] == 1].tolist() == [12, 12] and edges["alive"].sum() == 0

    # 2. chain with a short side branch joining back in (a tip): the tip goes, the chain stays
    graphs = _both()
    n0, _ = _chain(graphs, rng, 20)
    t0, _ = _chain(graphs, rng, 3, first_seq=100)
    for gph in graphs:
        gph.add_edge(t0 + 4, n0 + 2 * 10, 2500)          # tip end -> chain node 10 (forward)
        gph.add_edge(n0 + 2 * 10 + 1, t0 + 5, 2500)
        assert gph.run("tips") == 1
    layout.assert_same_graph(*graphs, "tip")
    nodes, _ = graphs[0].dump()
    assert nodes["alive"][t0: t0 + 6].sum() == 0 and nodes["alive"][n0: n0 + 40].all()

    # 3. a bubble: two parallel paths between chain nodes 5 and 9, the weaker one (fewer reads) goes
    graphs = _both()
    n0, _ = _chain(graphs, rng, 16)
    b0, _ = _chain(graphs, rng, 2, first_seq=200)
    for gph in graphs:
        gph.add_edge(n0 + 2 * 5, b0, 3100)
        gph.add_edge(b0 + 1, n0 + 2 * 5 + 1, 3100)
        gph.add_edge(b0 + 2, n0 + 2 * 9, 3100)
        gph.add_edge(n0 + 2 * 9 + 1, b0 + 3, 3100)
        assert gph.run("bubbles") >= 1
    layout.assert_same_graph(*graphs, "bubble")
    nodes, _ = graphs[0].dump()
    assert nodes["alive"][b0: b0 + 4].sum() == 0

    # 4. shrink: only chains of at least 2 * eps + 2 nodes are contracted, eps nodes stay at each end
    graphs = _both()
    _chain(graphs, rng, 30)
    for gph in graphs:
        assert gph.run("shrink", 20) == 0
        assert gph.run("shrink", 5) == 1
    layout.assert_same_graph(*graphs, "shrink")
    nodes, _ = graphs[0].dump()
    assert nodes["alive"].sum() == 2 * (5 + 5 + 1)


def _random_graph(graphs, seed):
    """a noisy assembly graph: a backbone chain, skip edges (transitive-looking shortcuts),
    short dead-end branches and parallel detours, all with consistent twins"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(30, 120))
    lens = rng.integers(4000, 12000, size=n)
    for gph in graphs:
        r = np.random.default_rng(seed)
        for k in range(n):
            gph.add_node_pair(k, b"r%d" % k, _random_dna(r, int(lens[k])))
    edges = []
    backbone = int(n * 0.7)
    for k in range(backbone - 1):
        if rng.random() < 0.03:
            continue                                  # a gap: two contigs
        edges.append((k, k + 1, int(rng.integers(500, 3500))))
    for _ in range(int(rng.integers(0, 8))):          # shortcuts over 2-3 nodes
        k = int(rng.integers(0, max(1, backbone - 4)))
        edges.append((k, k + int(rng.integers(2, 4)), int(rng.integers(3000, 3900))))
    extra = list(range(backbone, n))
    rng.shuffle(extra)
    while extra:
        m = min(len(extra), int(rng.integers(1, 5)))
        branch, extra = extra[:m], extra[m:]
        a = int(rng.integers(0, backbone))
        kind = rng.random()
        for x, y in zip(branch, branch[1:]):
            edges.append((x, y, int(rng.integers(500, 3500))))
        if kind < 0.4:                                # dead end hanging off the backbone
            edges.append((a, branch[0], int(rng.integers(500, 3500))))
        elif kind < 0.7:                              # branch that joins the backbone (a tip)
            edges.append((branch[-1], a, int(rng.integers(500, 3500))))
        else:                                         # detour around a stretch of the backbone
            b = min(backbone - 1, a + int(rng.integers(2, 7)))
            if b > a:
                edges.append((a, branch[0], int(rng.integers(500, 3500))))
                edges.append((branch[-1], b, int(rng.integers(500, 3500))))
    for gph in graphs:
        for a, b, l in edges:
            gph.add_edge(2 * a, 2 * b, l)
            gph.add_edge(2 * b + 1, 2 * a + 1, min(l, int(lens[b]) - 1))
    return n, len(edges)


@pytest.mark.parametrize("seed", range(60))
def test_random_graphs(seed):
    graphs = _both()
    _random_graph(graphs, seed)
    la, lb = [], []
    _simplify(graphs[0], la)
    _simplify(graphs[1], lb)
    assert la == lb
    layout.assert_same_graph(*graphs, "random graph %d" % seed)


def test_random_graphs_exercise_every_stage():
    tips = bubbles = shrunk = unitigs = 0
    for seed in range(60):
        g = layout.product()
        _random_graph((g,), seed)
        log = []
        _simplify(g, log)
        for x in log:
            if x[0] == "loop":
                tips += x[1]
                bubbles += x[2]
            elif x[0] == "round":
                tips += x[2]
            elif x[0] == "shrink":
                shrunk += x[1]
            else:
                unitigs += x[1]
    assert tips > 0 and bubbles > 0 and unitigs > 0, (tips, bubbles, shrunk, unitigs)


def _tangle(graphs, seed):
    """chains that cross in shared nodes (repeat-like junctions no tip / bubble rule resolves)
    plus a few shortcuts that are marked as transitive and removed: a component the layout runs on"""
    rng = np.random.default_rng(seed)
    n_chains, length = int(rng.integers(2, 4)), int(rng.integers(8, 16))
    ids = [[c * length + k for k in range(length)] for c in range(n_chains)]
    n = n_chains * length
    for gph in graphs:
        r = np.random.default_rng(seed)
        for k in range(n):
            gph.add_node_pair(k, b"r%d" % k, _random_dna(r, 9000))
    edges = []
    for ch in ids:
        for a, b in zip(ch, ch[1:]):
            edges.append((a, b, int(rng.integers(1000, 3000))))
    for c in range(1, n_chains):                       # cross links through the middle
        edges.append((ids[0][length // 2], ids[c][length // 2 + 1], 2500))
        edges.append((ids[c][length // 2 - 1], ids[0][length // 2], 2500))
    shortcuts = []
    for _ in range(4):
        c = int(rng.integers(0, n_chains)); k = int(rng.integers(0, length - 3))
        shortcuts.append(len(edges))
        edges.append((ids[c][k], ids[c][k + 2], 5000))
    for gph in graphs:
        for a, b, l in edges:
            gph.add_edge(2 * a, 2 * b, l)
            gph.add_edge(2 * b + 1, 2 * a + 1, l)
        for e in shortcuts:
            gph.mark_edge(2 * e)
        gph.note_transitive()
        gph.remove_marked(False)


@pytest.mark.parametrize("seed", range(12))
def test_layout_weights_and_long_edges(seed):
    """postprocess (force-directed layout, reference graph.cpp:1056-1279) and remove_long_edges:
    product (layout steps through the numpy stand-in of rala_hip_layout) against the oracle"""
    graphs = _both()
    _tangle(graphs, seed)
    for gph in graphs:
        gph.postprocess(seed)
    layout.assert_same_graph(*graphs, "after postprocess")
    w = graphs[0].edge_weights()
    assert (w > 0).any()
    la, lb = [], []
    _simplify(graphs[0], la)
    _simplify(graphs[1], lb)
    assert la == lb
    layout.assert_same_graph(*graphs, "after simplify")


@pytest.mark.parametrize("n,g,seed", [(3000, 600_000, 21), (2000, 1_000_000, 3)])
def test_on_disk_formats_match_oracle(n, g, seed):
    """print_csv / print_gfa / print_json (reference graph.cpp:2153-2297) field by field: the product's
    writers against the restatement, on the graph as built, after the clean-up stages (edge weights
    from the layout in the CSV) and after create_unitigs (unitig names in the GFA)"""
    import json

    prod, ora = _from_pipeline(n, g, seed)

    def check(tag):
        for kind in ("csv", "gfa", "json"):
            a, b = prod.print(kind), ora.print(kind)
            assert a == b, "%s %s differs at byte %d" % (tag, kind, next(i for i, (x, y) in enumerate(zip(a, b)) if x != y)
                                                         if len(a) == len(b) else -1)
        return prod.print("csv"), prod.print("gfa"), prod.print("json")

    csv, gfa, js = check("as built")
    nodes, edges = prod.dump()
    # CSV: one line per reverse-complement node with edges, one per edge: "id LN:i:len RC:i:reads,...,1,eid len weight"
    lines = csv.decode().splitlines()
    e_lines = [l for l in lines if l.split(",")[2] == "1"]
    assert len(e_lines) == int(edges["alive"].sum())
    eid, length, weight = e_lines[0].split(",")[3].split()
    assert int(length) == int(edges["length"][int(eid)]) and float(weight) == 0.0
    # GFA: S lines carry the sequence and LN / RC tags, L lines the overlap length <len(begin) - edge length>M
    g_lines = gfa.decode().splitlines()
    s_lines = [l.split("\t") for l in g_lines if l.startswith("S")]
    l_lines = [l.split("\t") for l in g_lines if l.startswith("L")]
    assert len(l_lines) == int(edges["alive"].sum()) and all(x[3].startswith("LN:i:") and int(x[3][5:]) == len(x[2]) for x in s_lines)
    assert all(x[2] in "+-" and x[4] in "+-" and x[5].endswith("M") for x in l_lines)
    # JSON: what misc/plotter.py walks - nodes{<read>: {n, p: [[read, node, rc, overlap]...], s: [...]}}, piles{}
    d = json.loads(js)
    assert set(d) <= {"nodes", "piles"}
    for key, v in d["nodes"].items():
        assert set(v) == {"n", "p", "s"} and all(len(x) == 4 for x in v["p"] + v["s"])
        assert key in d["piles"]
    # unitigs get names Utg<k> in the GFA (graph.cpp:2190-2196); chains between junctions collapse
    assert prod.run("unitigs") == ora.run("unitigs")
    csv1, gfa1, _ = check("after create_unitigs")
    assert b"\tUtg0\t" in gfa1 and csv1 != csv
    # layout rounds put weights on the edges (CSV column 3, third field)
    for gph in (prod, ora):
        gph.postprocess(0)
    csv2, _, _ = check("after postprocess")
    weights = [float(l.split(",")[3].split()[2]) for l in csv2.decode().splitlines() if l.split(",")[2] == "1"]
    assert weights and (max(weights) > 0 or seed == 21)       # (the 50x data set has nothing to lay out)
    _simplify(prod, [])
    _simplify(ora, [])
    check("after simplify")
