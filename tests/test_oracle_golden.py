"""CPU: the flat oracle (oracle/_build/liboracle.so) against the golden vectors that the
real reference objects produced."""
import numpy as np
import pytest

import golden_check as gc
from oracle.oracle import Oracle


@pytest.mark.parametrize("name", gc.SETS)
def test_flat_oracle_reproduces_golden(name):
    g = gc.load(name)
    ds = gc.dataset_for(g)
    n = ds.n_reads
    o = Oracle(ds.read_len, ds.overlaps, n_threads=4)
    assert o.backend == "flat-restatement"
    rc = o.initialize()
    assert rc == int(g["init_rc"])
    gc.same("valid", np.packbits(o.valid()), g["valid"])
    p = o.piles()
    for k in ("begin", "end", "median", "p10", "alive"):
        gc.same("p0_" + k, p[k], g["p0_" + k])
    dig = np.array([gc.data_digest(o.pile_data(r)) if p["alive"][r] else 0 for r in range(n)], dtype=np.uint64)
    gc.same("p0_data_digest", dig, g["p0_data_digest"])
    for kind, nm in ((0, "pits0"), (1, "hills0")):
        offs, flat = o.all_intervals(kind)
        gc.same(nm + "_off", offs, g[nm + "_off"])
        gc.same(nm, flat, g[nm])
    for r in g["data_reads"]:
        r = int(r)
        gc.same("data_%d" % r, o.pile_data(r), g["data_%d" % r])
        for q, tag in ((1.3, "130"), (1.82, "182"), (1.42, "142")):
            gc.same("slopes%s_%d" % (tag, r), o.find_slopes(r, q), g["slopes%s_%d" % (tag, r)])
    o.pass2()
    for which, nm in ((0, "p2_ov"), (1, "p2_int")):
        lst = o.overlap_list(which)
        for k, v in lst.items():
            gc.same("%s_%s" % (nm, k), v, g["%s_%s" % (nm, k)])
    gc.same("p2_alive", o.piles()["alive"], g["p2_alive"])
    hc = [o.hill_counts(r) for r in range(n)]
    gc.same("p2_hill_counts", np.concatenate(hc) if hc else np.zeros(0, np.uint32), g["p2_hill_counts"])
    o.preprocess_chimeras()
    p = o.piles()
    for k in ("begin", "end", "alive"):
        gc.same("p2f_" + k, p[k], g["p2f_" + k])
    for which, nm in ((0, "pp_ov"), (1, "pp_int")):
        lst = o.overlap_list(which)
        for k, v in lst.items():
            gc.same("%s_%s" % (nm, k), v, g["%s_%s" % (nm, k)])
    o.build_graph()
    gc.same("nodes", o.nodes(), g["nodes"])
    assert o.remove_transitive_edges() == int(g["n_tr"])
    for k, v in o.edges().items():
        gc.same("edge_" + k, v, g["edge_" + k])

    if "s_n" in g.files:
        o2 = Oracle(ds.read_len, ds.overlaps, n_threads=4)
        assert o2.initialize() == 0
        o2.pass2()
        o2.preprocess_chimeras()
        p = o2.piles()
        sens = ds.sensitive(p["alive"], p["begin"], p["end"])
        assert len(sens) == int(g["s_n"])
        o2.preprocess_repeats(sens)
        offs, flat = o2.all_intervals(2)
        gc.same("s_rep_off", offs, g["s_rep_off"])
        gc.same("s_rep", flat, g["s_rep"])
        fl = [o2.repeat_flags(r) for r in range(n)]
        gc.same("s_rep_flags", np.concatenate(fl) if fl else np.zeros(0, np.uint8), g["s_rep_flags"])
        p3 = o2.piles()
        gc.same("s_median", p3["median"], g["s_median"])
        gc.same("s_p10", p3["p10"], g["s_p10"])
        gc.same("s_ov_src", o2.overlap_list(0)["src"], g["s_ov_src"])
        o2.build_graph()
        assert o2.remove_transitive_edges() == int(g["s_n_tr"])
        for k, v in o2.edges().items():
            gc.same("s_edge_" + k, v, g["s_edge_" + k])


def test_crafted_parity_traps():
    """SURVEY Appendix B T1-T3 and T7 on a hand-made input (known answers from the reference objects)."""
    g, read_len, ov = gc.crafted_inputs()
    o = Oracle(read_len, ov, n_threads=1)
    o.pass1()
    gc.same("valid", o.valid(), g["valid"])
    # the closed forms the survey states
    assert o.valid().tolist() == [1, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 1]
    for r in range(len(read_len)):
        gc.same("data_%d" % r, o.pile_data(r), g["data_%d" % r])
    d4 = o.pile_data(4)     # b side of the span-20 overlap: bounds 45 (begin) and 35 (end)
    assert (d4[35:45] == 65535).all() and d4[34] == 0 and d4[45] == 0, "span < 30 must wrap the coverage (T1)"
    i = 0
    while "merge_in_%d" % i in g.files:
        gc.same("merge_%d" % i, o.interval_merge(g["merge_in_%d" % i]), g["merge_out_%d" % i])
        i += 1
    assert o.interval_merge([(10, 20), (30, 40), (18, 32)]).tolist() == [[10, 32], [10, 40]]
