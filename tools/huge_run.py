"""Round 6: one context with more than 2^30 overlaps (the partitioned bucketing's row offsets count bound pairs: the limit of
offsets in events, 4 n < 2^32, is gone up to 2^31 overlaps).  No oracle holds such a set; the checks are those of C5
(tests/test_gpu_fullsize.py::test_c5_properties): additivity over EVERY row (the sum of a row over its valid region = the sum of
the clipped, shrunk spans of the read's overlaps, Pile::add_layers, reference graph.cpp:311-326), nothing stored outside the
region, the run-space and the position-space pile kernels leaving the same rows (hash and sums of every row) and the same
annotations, a second transitive reduction finding nothing.

    python tools/huge_run.py [n_reads] [genome_len] [seed] [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rala_amd import hip
from rala_amd.synth import Dataset

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_500_000
genome = int(sys.argv[2]) if len(sys.argv) > 2 else 176_000_000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 6
out = sys.argv[4] if len(sys.argv) > 4 else None

t0 = time.time()
ds = Dataset(n_reads, genome, seed)
ov = ds.overlaps
res = {"n_reads": ds.n_reads, "n_overlaps": len(ov), "sum_read_len": int(ds.read_len.astype(np.int64).sum()),
       "beyond_2_30": len(ov) >= (1 << 30), "generated_s": round(time.time() - t0, 1)}
print("[huge] %d reads, %d overlaps (2^30 = %d), %.1f Gbase, generated in %.0f s" %
      (ds.n_reads, len(ov), 1 << 30, res["sum_read_len"] / 1e9, time.time() - t0), flush=True)


def one(**options):
    ctx = hip.Context(0)
    try:
        for k, v in options.items():
            ctx.set_option(k, v)
        ctx.set_reads(ds.read_len)
        ctx.set_overlaps(ov)
        t = time.time()
        ctx.initialize()
        wall = time.time() - t
        p = ctx.piles()
        rows = ctx.pile_row_digests()
        tm = dict(ctx.timings())
        r = dict(p=p, rows=rows, tm=tm, wall=wall, pits=ctx.intervals(0), hills=ctx.intervals(1))
        if not options:
            ctx.construct()
            r["n_tr"] = int(ctx.remove_transitive_edges())
            g = ctx.graph()
            keep = g["marked"] == 0
            marks, pairs = ctx.tr_mark(len(g["node_read"]), g["src"][keep], g["dst"][keep], g["len"][keep])
            r["second_tr"] = (int(pairs), bool(marks.any()))
            r["nodes"], r["edges"] = len(g["node_read"]), len(g["src"])
            r["tm_all"] = dict(ctx.timings())
        return r
    finally:
        ctx.close()


a = one()
p = a["p"]
print("[huge] initialize: %.1f ms on the device (bucketing %.1f, piles %.1f; %d reads beyond the first kernel's events, %d in position space), "
      "%d transitive pairs, %d nodes, %d edges" % (a["tm"]["total_ms"], a["tm"]["bucket_ms"], a["tm"]["pile_ms"], a["tm"]["pile_overflow_reads"],
                                                    a["tm"]["pile_position_reads"], a["n_tr"], a["nodes"], a["edges"]), flush=True)
fnv, inside, outside = a["rows"]
alive = p["alive"] != 0
assert not outside.any(), "something stored outside a valid region"
assert not fnv[~alive].any() and fnv[alive].all()
B, E = p["begin"].astype(np.int64), p["end"].astype(np.int64)
want = np.zeros(ds.n_reads, dtype=np.float64)
step = 1 << 26
t = time.time()
for lo_i in range(0, len(ov), step):
    sl = slice(lo_i, min(len(ov), lo_i + step))
    for side_id, sb, se in ((ov.a_id, ov.a_begin, ov.a_end), (ov.b_id, ov.b_begin, ov.b_end)):
        ids = side_id[sl].astype(np.int64)
        lo = np.clip(sb[sl].astype(np.int64) + 15, B[ids], E[ids])
        hi = np.clip(se[sl].astype(np.int64) - 15, B[ids], E[ids])
        want += np.bincount(ids, weights=np.maximum(hi - lo, 0).astype(np.float64), minlength=ds.n_reads)
bad = np.nonzero(alive & (inside.astype(np.float64) != want))[0]
assert len(bad) == 0, ("additivity", int(bad[0]), len(bad))
print("[huge] additivity over all %d live rows: ok (%.0f s on the host)" % (int(alive.sum()), time.time() - t), flush=True)
assert a["n_tr"] > 0 and a["second_tr"] == (0, False), (a["n_tr"], a["second_tr"])

b = one(use_run_kernel=0)
for k in ("begin", "end", "median", "p10", "alive"):
    assert (a["p"][k] == b["p"][k]).all(), k
for name in ("pits", "hills"):
    assert (a[name][0] == b[name][0]).all() and (a[name][1] == b[name][1]).all(), name
for k in range(3):
    assert (a["rows"][k] == b["rows"][k]).all(), ("rows", k)
print("[huge] position-space kernel: the same rows and annotations (%.1f ms)" % b["tm"]["total_ms"], flush=True)
if len(ov) < (1 << 30):          # (the other bucketing paths count events: below 2^30 overlaps they can be compared)
    c = one(use_partitioned_buckets=0)
    for k in range(3):
        assert (a["rows"][k] == c["rows"][k]).all(), ("rows, other bucketing", k)
    print("[huge] bucketing: partitioned %.2f ms, the single-pass path %.2f ms - the same rows" % (a["tm"]["bucket_ms"], c["tm"]["bucket_ms"]), flush=True)
    res.update(bucket_ms_partitioned=round(float(a["tm"]["bucket_ms"]), 3), bucket_ms_single_pass=round(float(c["tm"]["bucket_ms"]), 3))
res.update(n_alive=int(alive.sum()), transitive_pairs=a["n_tr"], nodes=a["nodes"], edges=a["edges"], second_tr_pairs=a["second_tr"][0],
           stage_ms={k: round(float(v), 3) for k, v in a["tm_all"].items()},
           initialize_ms=round(float(a["tm"]["total_ms"]), 2), initialize_ms_position_space=round(float(b["tm"]["total_ms"]), 2),
           rows_fnv_sum=int(fnv.sum(dtype=np.uint64)), rows_inside_sum=int(inside.sum(dtype=np.uint64)),
           checks=["additivity over every live row", "nothing outside the valid regions", "run-space == position-space kernel (rows, annotations)",
                   "second transitive reduction finds nothing"], ok=True)
print(json.dumps(res))
if out:
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
