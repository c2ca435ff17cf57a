"""Differential fuzzing of the HIP path against the oracle on random synthetic data sets
(python tests/fuzz_parity.py [n_cases] [first_seed]); prints one line per case, exits 1 on a
mismatch.  Besides size, coverage and plants a case may scale every coordinate (reads beyond the
position bitmap / beyond 65535 bases), shuffle the runs and add duplicates, unresolved names and
self overlaps, and run the sensitive pass (-s) on top."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from rala_amd import hip
from rala_amd.synth import Dataset
import parity
from rala_amd.cpus import effective_cpus
from test_gpu_parity import _Scaled, _shuffled_with_duplicates
from oracle.oracle import Oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for case in range(n_cases):
    seed = first + case
    rng = np.random.default_rng(seed)
    n = int(rng.integers(300, 5000)) * int(os.environ.get("RALA_FUZZ_SCALE", "1"))      # (RALA_FUZZ_SCALE=10: several partitions of the bucketing, tens of tiles)
    cov = float(rng.choice([8, 20, 35, 60, 110]))
    g = int(max(30_000, n * 10_000 / cov))
    plants = int(rng.integers(0, 16)) | (16 if rng.random() < 0.2 else 0)      # + heavy-tailed read lengths
    try:
        ds = Dataset(n, g, seed, plants)
        variant = []
        if rng.random() < 0.25:
            ds = _shuffled_with_duplicates(ds, seed)
            variant.append("shuffled")
        if rng.random() < 0.2:
            f = int(rng.choice([2, 3, 7]))
            ds = _Scaled(ds, f)
            variant.append("x%d" % f)
        sens_case = hasattr(ds, "sensitive") and rng.random() < 0.3
        st = parity.oracle_stages(ds, n_threads=effective_cpus())
        ctx = hip.Context(0)
        ctx.set_option("use_run_kernel", int(rng.random() < 0.85))
        ctx.set_option("use_fixed_buckets", int(rng.random() < 0.8))
        ctx.set_option("use_partitioned_buckets", int(rng.random() < 0.7))
        ctx.set_option("use_round_batches", int(rng.random() < 0.7))
        ctx.set_option("use_side_stream", int(rng.random() < 0.8))
        if rng.random() < 0.4:      # the containment fixed points' long lists' kernel / a mix of both
            ctx.set_option("debug_fp_lds_limit", int(rng.choice([0, 7, 100])))
        ctx.set_option("debug_part_shift", int(rng.choice([0, 0, 12, 13, 14])))     # first-level partitions of 4096 / 8192 / 16384 reads
        ctx.set_option("pile_chunk_mb", int(rng.choice([1024, 2, 2, 0])))          # the rows in chunks of 2 MB / in one hipMalloc
        ctx.set_option("debug_count_window", int(rng.choice([0, 0, 0, 3, 17])))    # the counting pass in windows of groups (beyond 4.9 M reads)
        ctx.set_option("debug_ev_events", int(rng.random() < 0.3))     # the rows' offsets in events (before round 6) / in pairs
        if rng.random() < 0.25:     # duplicate removal: a mark list that overflows (the pass over all overlaps takes over)
            ctx.set_option("debug_dedupe_list_cap", int(rng.choice([1, 3, 40])))
            variant.append("marks")
        # round 4: the position-space kernels' lists in global memory, from a few entries up (every read that reaches those
        # kernels); an interval pool that has to grow; the sensitive set handed over as device memory
        if rng.random() < 0.3:
            ctx.set_option("debug_force_big", 1)
            ctx.set_option("debug_big_caps", (int(rng.choice([4, 16, 1024])) << 32) | int(rng.choice([4, 16, 4096])))
            variant.append("big")
        if rng.random() < 0.2:
            ctx.set_option("interval_pool_per_read_x1000", int(rng.choice([0, 1, 50])))
            variant.append("pool")
        sens_dev = sens_case and rng.random() < 0.4
        if sens_dev:
            ctx.set_option("sensitive_in_device_memory", 1)
        ctx.set_reads(ds.read_len)
        ctx.set_overlaps(ds.overlaps)
        try:
            ctx.initialize()
            rc = 0
        except hip.RalaHipError as e:
            rc = e.code
        if st["init_rc"] != 0:
            assert rc == -4, (rc, st["init_rc"])
            print("case %d n=%d g=%d cov=%g plants=%d: all filtered (both)" % (seed, n, g, cov, plants), flush=True)
            continue
        assert rc == 0, rc
        parity.check_initialize(ctx, st, ds)
        if sens_case:
            variant.append("sensitive")
            ctx.set_option("use_gpu_tail", int(rng.random() < 0.7))
            o = Oracle(ds.read_len, ds.overlaps, n_threads=effective_cpus())
            assert o.initialize() == 0
            o.pass2()
            o.preprocess_chimeras()
            p = o.piles()
            sens = ds.sensitive(p["alive"], p["begin"], p["end"])
            o.preprocess_repeats(sens)
            want_rep, want_ov, want_p = o.all_intervals(2), o.overlap_list(0), o.piles()
            o.build_graph()
            want_tr, want_e = o.remove_transitive_edges(), o.edges()
            if sens_dev and len(sens):
                variant.append("dev")
                ctx.construct(hip.DeviceOverlaps.from_host(sens, 0))
            else:
                ctx.construct(sens)
            offs, pairs, flags = ctx.intervals(2)
            parity.assert_same("rep.offsets", offs, want_rep[0])
            parity.assert_same("rep.pairs", pairs, want_rep[1])
            hp = ctx.piles()
            for k in ("alive", "begin", "end", "median", "p10"):
                parity.assert_same("piles." + k, hp[k], want_p[k])
            want_rows = o.pile_row_digests()                # every row, the targets' with their second add_layers
            fnv, inside, _ = ctx.pile_row_digests()
            parity.assert_same("rows3.fnv", fnv, want_rows[0])
            parity.assert_same("rows3.sum", inside, want_rows[1])
            parity.assert_same("ov.src", ctx.overlap_list(0)["src"], want_ov["src"].astype(np.uint32))
            assert ctx.remove_transitive_edges() == want_tr
            gr = ctx.graph()
            for k in ("src", "dst", "len", "marked"):
                parity.assert_same("edges." + k, gr[k], want_e[k])
        else:
            ctx.construct()
            parity.check_construct(ctx, st)
            parity.check_tr(ctx, st)
        tm = ctx.timings()
        print("case %d n=%d g=%d cov=%g plants=%d %s: ok (%d overlaps, %d kept, overflow %d, position %d, lists in global memory %d, pool regrown %d)" % (
            seed, n, g, cov, plants, "+".join(variant) or "plain", len(ds.overlaps), len(st["ov"]["src"]),
            tm["pile_overflow_reads"], tm["pile_position_reads"], tm["pile_unbounded_reads"], tm["pool_regrown"]), flush=True)
        ctx.close()
    except AssertionError as e:
        bad += 1
        print("case %d n=%d g=%d cov=%g plants=%d: MISMATCH %s" % (seed, n, g, cov, plants, e), flush=True)
print("%d mismatches in %d cases" % (bad, n_cases))
sys.exit(1 if bad else 0)
