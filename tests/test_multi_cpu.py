"""CPU, gloo, world size 2 and 3: the Python side of the one-process-per-GPU launch
(rala_amd/multi.py) - the 128-byte id travels from rank 0 to everybody, all ranks cut the same
slices out of the overlap file and every record lands in exactly one of them, a rank that
fails to set up is noticed by all, the clock is the slowest rank's.  (The collectives of the
data path are C++ over RCCL and need GPUs: tests/test_gpu_sharded.py.)"""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from rala_amd import hip, multi

NO_READ = 0xFFFFFFFF


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _queries(seed, n_reads, n_ovl):
    rng = np.random.default_rng(seed)
    a = np.sort(rng.integers(0, n_reads, size=n_ovl)).astype(np.uint32)
    a[rng.random(n_ovl) < 0.03] = NO_READ            # names that do not resolve, inside runs too
    return a


def _worker(rank, world, port, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def make_id():
            calls.append(rank)
            return bytes((7 * i + seed) % 256 for i in range(128))
        uid = multi.exchange_id(make_id)
        a = _queries(seed, 53, 900)
        cuts = hip.slice_cuts(a, world)
        lo, hi = multi.my_slice(cuts, rank)
        ok_all = multi.all_agree(True)
        ok_one_bad = multi.all_agree(rank != world - 1)
        slowest = multi.max_over_ranks(10.0 + rank)
        q.put((rank, uid, calls, cuts, (lo, hi), ok_all, ok_one_bad, slowest))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rendezvous_slices_and_agreement(world):
    seed = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_id = bytes((7 * i + seed) % 256 for i in range(128))
    a = _queries(seed, 53, 900)
    covered = np.zeros(len(a), dtype=np.int64)
    for rank, uid, calls, cuts, (lo, hi), ok_all, ok_one_bad, slowest in res:
        assert uid == want_id                                   # the same 128 bytes everywhere
        assert calls == ([0] if rank == 0 else [])              # only rank 0 asked RCCL for an id
        assert cuts == res[0][3] and cuts[0] == 0 and cuts[-1] == len(a)
        covered[lo:hi] += 1
        assert ok_all is True and ok_one_bad is False           # one rank's failure is everybody's
        assert slowest == 10.0 + world - 1
    assert (covered == 1).all()                                 # every record in exactly one slice
    # no cut inside a run of equal queries; unresolved records do not break a run (graph.cpp:343-350)
    for c in res[0][3][1:-1]:
        if c == len(a) or a[c] == NO_READ:
            continue
        j = c
        while j > 0 and a[j - 1] == NO_READ:
            j -= 1
        assert j == 0 or a[j - 1] != a[c]


def test_slice_cuts_edge_cases():
    assert hip.slice_cuts(np.zeros(0, dtype=np.uint32), 4) == [0, 0, 0, 0, 0]
    assert hip.slice_cuts(np.full(10, 3, dtype=np.uint32), 3) == [0, 10, 10, 10]          # one run cannot be cut
    a = np.array([1, NO_READ, 1, NO_READ, NO_READ, 1, 2, 2], dtype=np.uint32)
    assert hip.slice_cuts(a, 2) == [0, 6, 8]            # X, <unresolved>, X stays together
    # a record whose TARGET does not resolve is skipped before the run logic as well (graph.cpp:338-350):
    # X b1 | Y <unknown> | X b1 is one run of X
    a = np.array([1, 1, 1, 2, 1, 1, 3, 3], dtype=np.uint32)
    b = np.array([5, 6, 7, NO_READ, 5, 8, 9, 9], dtype=np.uint32)
    assert hip.slice_cuts(a, 2, b) == [0, 6, 8]
    assert hip.slice_cuts(a, 2) == [0, 4, 8]            # (without the targets the Y record starts a run)
    a = np.arange(16, dtype=np.uint32)
    assert hip.slice_cuts(a, 4) == [0, 4, 8, 12, 16]
    assert hip.slice_cuts(a, 1) == [0, 16]


def _check_worker(rank, world, port, n_reads, q):
    """bench.py's result check over several processes: every rank hands in its slice of the validity bytes and the checksums of
    the rows it owns; rank 0 puts them back in file / read order and its verdict reaches everybody"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        valid = (np.arange(1000, dtype=np.uint32) * 7 % 5 != 0).astype(np.uint8)
        cuts = [1000 * k // world + (k % 2) for k in range(world)] + [1000]
        cuts[0] = 0
        fnv = np.arange(n_reads, dtype=np.uint64) * np.uint64(2654435761)
        tot = np.arange(n_reads, dtype=np.uint64) + np.uint64(11)
        mine = {"edges": "e", "n_tr": 3, "rank": rank} if rank in (0, world - 1) else None
        got = bench.gather_check_inputs(mine, valid[cuts[rank]:cuts[rank + 1]], fnv[rank::world], tot[rank::world], rank, world, n_reads)
        ok = None
        if rank == 0:
            first, last, v, f, t = got
            ok = (first["rank"] == 0 and last["rank"] == world - 1 and (v == valid).all() and (f == fnv).all() and (t == tot).all())
        else:
            assert got is None
        verdict = bench.broadcast_flag(bool(ok))
        verdict_bad = bench.broadcast_flag(False if rank == 0 else True)       # rank 0's word counts
        q.put((rank, verdict, verdict_bad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_reads", [(2, 101), (3, 100)])
def test_result_check_inputs_over_ranks(world, n_reads):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_check_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] is True and r[2] is False for r in res)
