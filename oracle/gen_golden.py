#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates the golden vectors under tests/golden/ by running the hot path on the REAL
reference objects (oracle/_ref/liboracle_ref.so: rala::Pile and rala::Overlap compiled
from /root/reference/src/pile.cpp and overlap.cpp; orchestration restated in
oracle_graph.hpp).  Runs only where /root/reference exists; the fixtures are data
(inputs are regenerated from the seed by rala_amd/synth, outputs are stored here).

    python oracle/gen_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.oracle import Oracle, have_ref, build  # noqa: E402
from rala_amd.synth import Dataset, Overlaps  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# name -> (n_reads, genome, seed, plants, with_sensitive)
SETS = {
    "c1": (1000, 200_000, 1, 15, True),            # BASELINE configs[0]
    "sparse": (1500, 1_500_000, 42, 15, False),    # ~10x coverage: reads die in find_valid_region
    "plain": (800, 160_000, 5, 0, False),          # no planted artefacts
    "dense": (600, 60_000, 9, 15, True),           # ~100x coverage, many duplicates
}


def data_digest(a):
    """first 8 bytes of the SHA-256 of a coverage vector, as uint64"""
    return int.from_bytes(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()[:8], "little")


def input_digest(ds):
    m = hashlib.sha256()
    m.update(ds.read_len.tobytes())
    for a in ds.overlaps.arrays():
        m.update(a.tobytes())
    return m.hexdigest()


def crafted():
    """Hand-made case for the parity traps of SURVEY Appendix B (T1-T3): duplicates with
    strictly greatest / tied lengths, a self overlap, an unknown name, separate runs."""
    read_len = np.array([4000, 4200, 3900, 4100, 5000], dtype=np.uint32)
    rows = [
        # a, b, ab, ae, bb, be, len, strand
        (0, 1, 100, 1100, 50, 1050, 1000, 0),
        (0, 1, 100, 1000, 50, 950, 900, 0),        # shorter duplicate -> invalid
        (0, 2, 200, 1200, 0, 1000, 1000, 1),
        (0, 2, 210, 1210, 0, 1000, 1000, 1),        # tie -> the LAST stays
        (0, 0, 100, 600, 2000, 2500, 500, 0),       # self overlap -> invalid, bounds twice
        (1, 0xFFFFFFFF, 0, 500, 0, 500, 500, 0),    # unknown name -> invalid, no bounds, run not broken
        (1, 3, 0, 2000, 100, 2100, 2000, 0),
        (1, 3, 5, 2100, 100, 2195, 2095, 0),        # longer, later -> first invalid
        (1, 3, 0, 1500, 100, 1600, 1500, 0),
        (2, 1, 100, 3000, 200, 3100, 2900, 0),
        (1, 3, 0, 2000, 100, 2100, 2100, 0),        # new run of a=1: not deduped against the first
        (3, 4, 1000, 1020, 30, 50, 20, 0),          # span < 30: coverage wraps to 65535 (T1)
    ]
    a = np.array(rows, dtype=np.uint64)
    ov = Overlaps(a_id=a[:, 0], b_id=a[:, 1], a_begin=a[:, 2], a_end=a[:, 3], b_begin=a[:, 4], b_end=a[:, 5],
                  length=a[:, 6], strand=a[:, 7])
    return read_len, ov


def run_set(name, n, g, seed, plants, with_sens):
    ds = Dataset(n, g, seed, plants)
    o = Oracle(ds.read_len, ds.overlaps, n_threads=8, ref=True)
    assert o.backend == "reference-objects"
    out = {"params": np.array([n, g, seed, plants], dtype=np.uint64),
           "input_sha256": np.frombuffer(bytes.fromhex(input_digest(ds)), dtype=np.uint8),
           "n_overlaps": np.uint64(len(ds.overlaps))}
    rc = o.initialize()
    out["init_rc"] = np.int64(rc)
    out["valid"] = np.packbits(o.valid())
    p = o.piles()
    for k, v in p.items():
        out["p0_" + k] = v
    alive = np.nonzero(p["alive"])[0]
    out["p0_data_digest"] = np.array([data_digest(o.pile_data(r)) if p["alive"][r] else 0 for r in range(n)],
                                  dtype=np.uint64)
    for kind, nm in ((0, "pits0"), (1, "hills0")):
        offs, flat = o.all_intervals(kind)
        out[nm + "_off"] = offs
        out[nm] = flat
    # full coverage vectors + raw slope regions for the interesting piles
    special = sorted(set(np.nonzero(np.diff(out["pits0_off"].astype(np.int64)))[0].tolist()[:6] +
                         np.nonzero(np.diff(out["hills0_off"].astype(np.int64)))[0].tolist()[:6] +
                         alive[:3].tolist()))
    out["data_reads"] = np.array(special, dtype=np.uint32)
    for r in special:
        out["data_%d" % r] = o.pile_data(r)
        for q, tag in ((1.3, "130"), (1.82, "182"), (1.42, "142")):
            out["slopes%s_%d" % (tag, r)] = o.find_slopes(r, q)
    if rc != 0:
        return out, ds
    o.pass2()
    for which, nm in ((0, "p2_ov"), (1, "p2_int")):
        lst = o.overlap_list(which)
        for k, v in lst.items():
            out["%s_%s" % (nm, k)] = v
    out["p2_alive"] = o.piles()["alive"]
    hc = [o.hill_counts(r) for r in range(n)]
    out["p2_hill_counts"] = np.concatenate(hc) if hc else np.zeros(0, np.uint32)
    o.preprocess_chimeras()
    p = o.piles()
    for k in ("begin", "end", "alive"):
        out["p2f_" + k] = p[k]
    for which, nm in ((0, "pp_ov"), (1, "pp_int")):
        lst = o.overlap_list(which)
        for k, v in lst.items():
            out["%s_%s" % (nm, k)] = v
    # the graph without the sensitive pass
    o.build_graph()
    out["nodes"] = o.nodes()
    out["n_tr"] = np.uint64(o.remove_transitive_edges())
    for k, v in o.edges().items():
        out["edge_" + k] = v
    if with_sens:
        # second run with the sensitive overlap set derived from the survivors
        o2 = Oracle(ds.read_len, ds.overlaps, n_threads=8, ref=True)
        assert o2.initialize() == 0
        o2.pass2()
        o2.preprocess_chimeras()
        p = o2.piles()
        sens = ds.sensitive(p["alive"], p["begin"], p["end"])
        out["s_n"] = np.uint64(len(sens))
        o2.preprocess_repeats(sens)
        offs, flat = o2.all_intervals(2)
        out["s_rep_off"] = offs
        out["s_rep"] = flat
        fl = [o2.repeat_flags(r) for r in range(n)]
        out["s_rep_flags"] = np.concatenate(fl) if fl else np.zeros(0, np.uint8)
        p3 = o2.piles()
        out["s_median"] = p3["median"]
        out["s_p10"] = p3["p10"]
        lst = o2.overlap_list(0)
        out["s_ov_src"] = lst["src"]
        o2.build_graph()
        out["s_n_tr"] = np.uint64(o2.remove_transitive_edges())
        for k, v in o2.edges().items():
            out["s_edge_" + k] = v
    return out, ds


def run_crafted():
    read_len, ov = crafted()
    o = Oracle(read_len, ov, n_threads=1, ref=True)
    o.pass1()
    out = {"read_len": read_len}
    for nm, a in zip(("a_id", "b_id", "a_begin", "a_end", "b_begin", "b_end", "length", "strand"), ov.arrays()):
        out["in_" + nm] = a
    out["valid"] = o.valid()
    for r in range(len(read_len)):
        out["data_%d" % r] = o.pile_data(r)
    # intervalMerge known answers (B-T7 example included)
    cases = [[(10, 20), (30, 40), (18, 32)], [(5, 9), (1, 6), (8, 12), (20, 30)], [(1, 2)], [],
             [(100, 200), (150, 160), (190, 300), (10, 120)]]
    for i, c in enumerate(cases):
        a = np.array(c, dtype=np.uint32).reshape(-1, 2)
        out["merge_in_%d" % i] = a
        out["merge_out_%d" % i] = o.interval_merge(a)
    return out


def main():
    if not have_ref():
        build()
    if not have_ref():
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    for name, (n, g, seed, plants, with_sens) in SETS.items():
        out, ds = run_set(name, n, g, seed, plants, with_sens)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print("%s: %d reads, %d overlaps -> %s (%d KiB)" % (name, n, len(ds.overlaps), path,
                                                            os.path.getsize(path) // 1024))
    path = os.path.join(OUT, "crafted.npz")
    np.savez_compressed(path, **run_crafted())
    print("crafted ->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
