"""GPU: EVERY pile row against the oracle (round 6; VERDICT round 5 "the 20 GB of pile rows are checked on a sample").

rala_hip_get_pile_row_digests hashes every row where it lies in device memory - per read the FNV-1a-64 over the bytes of
Pile::data() (reference src/pile.hpp:53) - and the oracle computes the same vector from its objects (ora_pile_row_digests).
Here at the sizes the oracle runs in seconds, with every kernel variant that writes rows; at the full BASELINE sizes the digest
of the whole vector is compared with tests/golden/fullsize_*.json (test_gpu_fullsize.py)."""
import numpy as np
import pytest

from rala_amd.synth import Dataset

import parity
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


def fnv1a(data):
    h = 1469598103934665603
    for b in np.ascontiguousarray(data, dtype=np.uint16).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def check_rows(ctx, o, after_initialize):
    want_fnv, want_sum = o.pile_row_digests()
    fnv, inside, outside = ctx.pile_row_digests()
    parity.assert_same("row fnv", fnv, want_fnv)
    parity.assert_same("row sum", inside, want_sum)
    if after_initialize:
        assert not outside.any()            # Pile::shrink zeroed what lies outside the region, and the row was written once
    return fnv


@pytest.mark.parametrize("options", [{}, {"use_run_kernel": 0}, {"use_partitioned_buckets": 0}, {"use_fixed_buckets": 0},
                                     {"debug_force_big": 1, "debug_big_caps": (4 << 32) | 4}])
@pytest.mark.parametrize("n,g,seed,plants", [(3000, 600_000, 21, 15), (2000, 1_200_000, 33, 31), (600, 60_000, 9, 15)])
def test_every_row_matches_the_oracle(hip_ctx_factory, n, g, seed, plants, options):
    ds = Dataset(n, g, seed, plants)
    o = ora.Oracle(ds.read_len, ds.overlaps, n_threads=8, ref=ora.have_ref())
    assert o.initialize() == 0
    ctx = hip_ctx_factory()
    for k, v in options.items():
        ctx.set_option(k, v)
    ctx.set_reads(ds.read_len)
    ctx.set_overlaps(ds.overlaps)
    ctx.initialize()
    fnv = check_rows(ctx, o, True)
    # the device's hash is the hash of what the getter hands out
    alive = np.nonzero(ctx.piles()["alive"])[0]
    for r in alive[:: max(1, len(alive) // 12)]:
        assert int(fnv[r]) == fnv1a(ctx.pile_data(int(r))), int(r)
    assert not fnv[ctx.piles()["alive"] == 0].any()
    # the chimera stage narrows regions (Pile::shrink zeroes outside them); rows are not rewritten, the regions apply
    o.pass2()
    o.preprocess_chimeras()
    ctx.construct()
    check_rows(ctx, o, False)


def test_every_row_long_reads_and_dense_reads(hip_ctx_factory):
    """the other instantiations of the pile kernel: reads beyond 16384 / 32768 bases, reads with more events than 512 / 1024"""
    from test_gpu_parity import _Scaled

    for ds in (_Scaled(Dataset(1500, 300_000, 11), 3), _Scaled(Dataset(800, 160_000, 5), 7), Dataset(1200, 24_000, 3)):
        o = ora.Oracle(ds.read_len, ds.overlaps, n_threads=8, ref=ora.have_ref())
        assert o.initialize() == 0
        ctx = hip_ctx_factory()
        ctx.set_reads(ds.read_len)
        ctx.set_overlaps(ds.overlaps)
        ctx.initialize()
        check_rows(ctx, o, True)
        ctx.close()


def test_every_row_wrapped_coverage(hip_ctx_factory):
    """rows that hold (0 - k) mod 2^16 (tests/wrapcase.py)"""
    import wrapcase

    read_len, ov, _kinds = wrapcase.wrap_inputs(seed=1)
    o = ora.Oracle(read_len, ov, n_threads=4, ref=ora.have_ref())
    assert o.initialize() == 0
    for run_kernel in (1, 0):
        ctx = hip_ctx_factory()
        ctx.set_option("use_run_kernel", run_kernel)
        ctx.set_reads(read_len)
        ctx.set_overlaps(ov)
        ctx.initialize()
        check_rows(ctx, o, True)


@pytest.mark.parametrize("world", [3, 8])
def test_every_row_of_a_sharded_run(world):
    """the owners' rows (rala_hip_mg_get_pile_row_digests: entry j of rank k = read j * world + k) under the final regions"""
    from test_gpu_sharded import Sharded

    ds = Dataset(3000, 600_000, 21, 31)
    o = ora.Oracle(ds.read_len, ds.overlaps, n_threads=8, ref=ora.have_ref())
    assert o.construct() == 0
    want_fnv, want_sum = o.pile_row_digests()
    sh = Sharded(ds, world)
    try:
        sh.run()
        fnv = np.zeros(ds.n_reads, dtype=np.uint64)
        tot = np.zeros(ds.n_reads, dtype=np.uint64)
        for k, r in enumerate(sh.ranks):
            f, s, _ = r.pile_row_digests()
            fnv[k::world] = f
            tot[k::world] = s
        parity.assert_same("row fnv", fnv, want_fnv)
        parity.assert_same("row sum", tot, want_sum)
    finally:
        sh.close()
