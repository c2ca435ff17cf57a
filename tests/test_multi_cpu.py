"""CPU: the collective plumbing of rala_amd/multi.py under gloo with world_size 2 and 3:
owner split + variable all-to-all + padded all-gather + global re-indexing reproduce what a
single process computes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rala_amd import multi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_case(seed, n_reads, n_ovl):
    rng = np.random.default_rng(seed)
    a = np.sort(rng.integers(0, n_reads, size=n_ovl)).astype(np.int64)
    b = rng.integers(0, n_reads, size=n_ovl).astype(np.int64)
    b[rng.random(n_ovl) < 0.02] = multi.NO_READ          # unresolved names
    a_unres = rng.random(n_ovl) < 0.01
    bounds = rng.integers(0, 2 ** 32, size=(n_ovl, 4), dtype=np.uint64).astype(np.int64)
    reads = np.stack([a, a, b, b], axis=1)
    bad = (b == multi.NO_READ) | a_unres
    reads[bad] = multi.NO_READ
    return a, reads.reshape(-1), bounds.reshape(-1)


def _worker(rank, world, port, seed, n_reads, n_ovl, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, reads, bounds = _make_case(seed, n_reads, n_ovl)
        cuts = multi.slice_starts(a, world)
        lo, hi = 4 * cuts[rank], 4 * cuts[rank + 1]
        lr, bd, counts = multi.owner_split(torch.from_numpy(reads[lo:hi]), torch.from_numpy(bounds[lo:hi]), world)
        lr2, _ = multi.all_to_all_v(lr.to(torch.int32), counts)
        bd2, _ = multi.all_to_all_v(bd.to(torch.int32), counts)
        # per local read: multiset of received bounds
        got = sorted(zip(lr2.tolist(), (bd2.to(torch.int64) & 0xFFFFFFFF).tolist()))
        # a per-read annotation computed by the owner, then gathered + interleaved
        nl = multi.n_local_reads(n_reads, rank, world)
        cnt = np.bincount(lr2.numpy(), minlength=nl)[:nl].astype(np.int64)
        parts = [x.numpy() for x in multi.all_gather_v(torch.from_numpy(cnt))]
        glob = multi.interleave(parts, n_reads, world)
        # intervals: local read j of rank k gets (j % 3) rows [global id, i]
        rows = [[j * world + rank, i] for j in range(nl) for i in range(j % 3)]
        ic = np.array([j % 3 for j in range(nl)], dtype=np.int64)
        fl = np.array(rows, dtype=np.int64).reshape(-1)
        pc = [x.numpy() for x in multi.all_gather_v(torch.from_numpy(ic))]
        pf = [x.numpy() for x in multi.all_gather_v(torch.from_numpy(fl))]
        offs, flat = multi.merge_intervals(pc, pf, n_reads, world, 2)
        q.put((rank, got, glob.tolist(), offs.tolist(), flat.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tuple_routing_and_gather(world):
    seed, n_reads, n_ovl = 11, 37, 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, n_reads, n_ovl, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, got, glob, offs, flat = q.get(timeout=120)
        res[rank] = (got, glob, offs, flat)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, reads, bounds = _make_case(seed, n_reads, n_ovl)
    keep = reads != multi.NO_READ
    want_cnt = np.bincount(reads[keep], minlength=n_reads)
    for rank in range(world):
        got, glob, offs, flat = res[rank]
        mine = keep & (reads % world == rank)
        want = sorted(zip((reads[mine] // world).tolist(), (bounds[mine] & 0xFFFFFFFF).tolist()))
        assert got == want
        assert glob == want_cnt.tolist()
        # intervals come back in global read order with their rows in order
        exp_rows = [[r, i] for r in range(n_reads) for i in range((r // world) % 3)]
        assert flat == exp_rows
        assert offs[-1] == len(exp_rows)


def test_slice_starts_on_run_boundaries():
    a = np.array([0, 0, 0, 1, 1, 2, 2, 2, 2, 3, 5, 5, 7])
    for world in (1, 2, 3, 4, 8):
        cuts = multi.slice_starts(a, world)
        assert cuts[0] == 0 and cuts[-1] == len(a) and len(cuts) == world + 1
        assert all(x <= y for x, y in zip(cuts, cuts[1:]))
        for c in cuts[1:-1]:
            assert c == len(a) or c == 0 or a[c] != a[c - 1]


def _local_state(rank, world, n_reads, seed):
    """what an owner rank would hold after initialize: per-read fields + an interval pool"""
    rng = np.random.default_rng(seed * 100 + rank)
    n = multi.n_local_reads(n_reads, rank, world)
    f = {
        "begin": rng.integers(0, 1 << 20, n).astype(np.uint32), "end": rng.integers(0, 1 << 31, n).astype(np.uint32) * 2,
        "median": rng.integers(0, 1 << 16, n).astype(np.uint16), "p10": rng.integers(0, 1 << 16, n).astype(np.uint16),
        "alive": rng.integers(0, 2, n).astype(np.uint8), "n_pits": rng.integers(0, 3, n).astype(np.uint8),
        "n_hills": rng.integers(0, 3, n).astype(np.uint8),
    }
    cnt = f["n_pits"].astype(np.int64) + f["n_hills"]
    slot = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    order = rng.permutation(n)                         # pool order is arbitrary (atomic allocation)
    pos = 0
    pool = []
    for j in order:
        if cnt[j]:
            slot[j] = pos
            for i in range(cnt[j]):
                pool.append((j * world + rank, i, 7 * i + rank))
            pos += cnt[j]
    f["slot"] = slot
    return f, np.array(pool, dtype=np.uint32).reshape(-1, 3)


def _state_worker(rank, world, port, seed, n_reads, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f, pool = _local_state(rank, world, n_reads, seed)
        nl = multi.padded_local(n_reads, world)
        off, total = multi.state_layout(nl)
        packed = np.zeros(total, dtype=np.uint8)
        for name, w in multi.STATE_FIELDS:
            raw = f[name].view(np.uint8)
            packed[off[name]: off[name] + raw.size] = raw
        rows = multi.all_gather_rows(torch.from_numpy(packed))
        pools, counts = multi.gather_pools(torch.from_numpy(pool.reshape(-1).view(np.uint8).copy()))
        st = multi.unpack_state(rows, nl, n_reads, counts)
        q.put((rank, {k: v.numpy().tolist() for k, v in st.items()},
               pools.numpy().view(np.uint32).reshape(-1, 3).tolist(), counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_reads", [(2, 37), (3, 41), (2, 1)])
def test_packed_state_gather(world, n_reads):
    seed = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_state_worker, args=(r, world, port, seed, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    locs = [_local_state(r, world, n_reads, seed) for r in range(world)]
    for rank, st, pools, counts in res:
        assert counts == [len(l[1]) for l in locs]
        for name, w in multi.STATE_FIELDS:
            if name == "slot":
                continue
            want = multi.interleave([l[0][name] for l in locs], n_reads, world)
            got = np.array(st[name], dtype=np.int64) & ((1 << (8 * w)) - 1)
            assert got.tolist() == want.astype(np.int64).tolist(), name
        # every read's intervals are found through its rebased slot, in order
        for r in range(n_reads):
            k, j = r % world, r // world
            cnt = int(locs[k][0]["n_pits"][j]) + int(locs[k][0]["n_hills"][j])
            s = st["slot"][r]
            if cnt == 0:
                assert s == multi.NO_SLOT
            else:
                assert [tuple(x) for x in pools[s: s + cnt]] == [(r, i, 7 * i + k) for i in range(cnt)]
