"""Generates tests/golden/fullsize_<workload>.json: SHA-256 digests of every stage of the ORACLE on a
full-size BASELINE workload (c2 = 100 k reads / 5 M overlaps, c3 = 1 M reads / 50 M overlaps), so
that the GPU suite can check the HIP path at full size without running the oracle there.

    python tests/golden/make_fullsize_digests.py c2 [threads]
    python tests/golden/make_fullsize_digests.py c2 [threads] sens      -> fullsize_c2_sens.json: the same
        after the sensitive pass (-s) with the generator's sensitive overlap set
    ... flat                                                            -> flat restatement instead

The oracle runs on the reference's own rala::Pile / rala::Overlap objects (oracle/_ref, compiled
from /root/reference/src/pile.cpp and overlap.cpp where they lie) whenever that library is built;
the JSON records which backend produced it ("backend").  c5x = 200 k reads at C5's 75x coverage
(BASELINE configs[4] itself, 4 M reads / 300 M overlaps, does not fit this container's memory).

Needs roughly 2.5 GB (c2) / 8 GB (c5x) / 40 GB (c3) of host memory."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from rala_amd.synth import Dataset
from oracle.oracle import Oracle, have_ref

SAMPLE = 400          # reads whose whole coverage vector is digested


def dg(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def sample_reads(n_reads, alive):
    rng = np.random.default_rng(12345)
    live = np.nonzero(alive)[0]
    return np.sort(rng.choice(live, size=min(SAMPLE, len(live)), replace=False))


def main():
    wl = sys.argv[1]
    threads = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else (os.cpu_count() or 1)
    with_sens = "sens" in sys.argv[2:]
    use_ref = have_ref() and "flat" not in sys.argv[2:]
    t0 = time.time()
    ds = Dataset.config(wl)
    o = Oracle(ds.read_len, ds.overlaps, n_threads=threads, ref=use_ref)
    out = {"workload": wl, "backend": o.backend, "n_reads": int(ds.n_reads), "n_overlaps": len(ds.overlaps)}
    assert o.initialize() == 0
    p = o.piles()
    out["valid"] = dg(np.packbits(o.valid()))
    out["piles0"] = dg(*[p[k] for k in ("begin", "end", "median", "p10", "alive")])
    pits, hills = o.all_intervals(0), o.all_intervals(1)
    out["pits0"] = dg(pits[0].astype(np.uint64), pits[1].astype(np.uint32))
    out["hills0"] = dg(hills[0].astype(np.uint64), hills[1].astype(np.uint32))
    reads = sample_reads(ds.n_reads, p["alive"])
    out["data_reads"] = dg(reads.astype(np.int64))
    out["data0"] = dg(*[np.asarray(o.pile_data(int(r)), dtype=np.uint16) for r in reads])
    # EVERY pile row (round 6): the vector of per-read FNV-1a-64 over the bytes of data_ of all live reads (0 for a filtered read)
    # and the vector of the rows' sums - the HIP path computes the same two vectors where the rows lie (rala_hip_get_pile_row_digests)
    fnv, tot = o.pile_row_digests()
    out["rows0"] = dg(fnv)
    out["rows0_sum"] = dg(tot)
    print("[digest] initialize done %.0f s" % (time.time() - t0), file=sys.stderr)
    o.pass2()
    o.preprocess_chimeras()
    p2 = o.piles()
    out["piles2"] = dg(p2["begin"], p2["end"], p2["alive"])
    fnv, tot = o.pile_row_digests()         # (the chimera stage narrows regions: Pile::shrink zeroes what falls outside)
    out["rows2"] = dg(fnv)
    out["rows2_sum"] = dg(tot)
    ov, it = o.overlap_list(0), o.overlap_list(1)
    out["n_overlaps_kept"] = int(len(ov["src"]))
    out["n_internals_kept"] = int(len(it["src"]))
    out["ov"] = dg(*[np.asarray(ov[k]).astype(np.uint32) for k in ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")])
    out["int"] = dg(*[np.asarray(it[k]).astype(np.uint32) for k in ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")])
    if with_sens:
        # Graph::preprocess(overlaps, sensitive overlaps): the set derived from the piles as they are now
        sens = ds.sensitive(p2["alive"], p2["begin"], p2["end"])
        out["n_sensitive"] = len(sens)
        o.preprocess_repeats(sens)
        rep = o.all_intervals(2)
        flags = [o.repeat_flags(int(r)) for r in np.nonzero(np.diff(rep[0]))[0]]
        out["rep"] = dg(rep[0].astype(np.uint64), rep[1].astype(np.uint32),
                        (np.concatenate(flags) if flags else np.zeros(0, np.uint8)).astype(np.uint8))
        out["n_repeat_hills"] = int(len(rep[1]))
        p3 = o.piles()
        out["piles3"] = dg(*[p3[k] for k in ("begin", "end", "median", "p10", "alive")])
        fnv, tot = o.pile_row_digests()     # (the targets' rows hold the second add_layers now)
        out["rows3"] = dg(fnv)
        out["rows3_sum"] = dg(tot)
        targets = np.unique(sens.b_id)[:SAMPLE]
        out["data3"] = dg(*[np.asarray(o.pile_data(int(r)), dtype=np.uint16) for r in targets])
        ov = o.overlap_list(0)
        out["n_overlaps_kept_sens"] = int(len(ov["src"]))
        out["ov_sens"] = dg(*[np.asarray(ov[k]).astype(np.uint32) for k in ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")])
        print("[digest] sensitive pass done %.0f s" % (time.time() - t0), file=sys.stderr)
    o.build_graph()
    out["nodes"] = dg(o.nodes().astype(np.uint32))
    out["n_tr"] = int(o.remove_transitive_edges())
    e = o.edges()
    out["n_edges"] = int(len(e["src"]))
    out["edges"] = dg(e["src"].astype(np.uint32), e["dst"].astype(np.uint32), e["len"].astype(np.uint32), e["marked"].astype(np.uint8))
    out["oracle_seconds"] = round(time.time() - t0, 1)
    path = os.path.join(ROOT, "tests", "golden", "fullsize_%s%s.json" % (wl, "_sens" if with_sens else ""))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
