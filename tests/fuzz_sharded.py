"""Differential fuzzing of the SHARDED runner (ranks as threads on one GPU, in-process transport) against the oracle:
python tests/fuzz_sharded.py [n_cases] [first_seed].  Random size (from a dozen reads), coverage, plants, world 2 .. 8,
run order shuffled with duplicates / unresolved names, tuples instead of bound records, the containment fixed points
through the long lists' kernel.  One line per case, exits 1 on a mismatch."""
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
from test_gpu_parity import _shuffled_with_duplicates
from test_gpu_sharded import Sharded, check_rank

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
bad = 0
for case in range(n_cases):
    seed = first + case
    rng = np.random.default_rng(seed)
    n = int(rng.choice([12, 30, 70, 200, 800, 2500]) * (1 + rng.random()))
    cov = float(rng.choice([8, 20, 35, 60]))
    g = int(max(12_000, n * 10_000 / cov))
    plants = int(rng.integers(0, 16)) | (16 if rng.random() < 0.15 else 0)
    world = int(rng.integers(2, 9))
    variant = []
    sh = None
    try:
        ds = Dataset(n, g, seed, plants)
        if rng.random() < 0.25:
            ds = _shuffled_with_duplicates(ds, seed)
            variant.append("shuffled")
        st = parity.oracle_stages(ds, n_threads=effective_cpus())
        sh = Sharded(ds, world)
        opts = {}
        form = rng.random()              # (default: the bounds scattered once, by (owner, partition), on the senders)
        if form < 0.15:
            opts["use_fused_emit"] = 0
            opts["use_bound_records"] = 0
            variant.append("tuples")
        elif form < 0.35:
            opts["use_fused_emit"] = 0
            variant.append("records")
        if rng.random() < 0.4:
            opts["debug_fp_lds_limit"] = int(rng.choice([0, 7, 100]))
            variant.append("limit%d" % opts["debug_fp_lds_limit"])
        for r in sh.ranks:
            for k, v in opts.items():
                r.context().set_option(k, v)
        what = "world %d n=%d g=%d plants=%d %s" % (world, n, g, plants, "+".join(variant) or "plain")
        if st["init_rc"] != 0:
            try:
                sh.run()
                raise AssertionError("the oracle filtered everything, the ranks did not")
            except hip.RalaHipError as e:
                assert e.code == -4, e
            print("case %d %s: all filtered (both)" % (seed, what), flush=True)
            continue
        n_tr = sh.run()
        for r in sh.ranks:
            check_rank(r.context(), st, n_tr)
        # every row from its owner, under the final regions (round 6): the oracle's objects hold the same after the chimera stage
        want_fnv, want_sum = st["oracle"].pile_row_digests()
        fnv = np.zeros(ds.n_reads, dtype=np.uint64)
        tot = np.zeros(ds.n_reads, dtype=np.uint64)
        for k, r in enumerate(sh.ranks):
            fnv[k::world], tot[k::world], _ = r.pile_row_digests()
        parity.assert_same("rows2.fnv", fnv, want_fnv)
        parity.assert_same("rows2.sum", tot, want_sum)
        assert sh.run() == n_tr                 # the same objects once more
        check_rank(sh.ranks[world - 1].context(), st, n_tr)
        print("case %d %s: ok (%d overlaps, %d kept, %d pairs)" % (seed, what, len(ds.overlaps), len(st["ov"]["src"]), n_tr), flush=True)
    except Exception as e:                     # noqa: BLE001 - a fuzzer reports and goes on
        bad += 1
        print("case %d world %d n=%d g=%d plants=%d %s: MISMATCH %s: %s" % (seed, world, n, g, plants, "+".join(variant), type(e).__name__, str(e)[:300]), flush=True)
    finally:
        if sh is not None:
            sh.close()
print("%d mismatches in %d cases" % (bad, n_cases))
sys.exit(1 if bad else 0)
