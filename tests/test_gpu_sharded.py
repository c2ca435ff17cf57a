"""GPU (one device): the multi-GPU decomposition of Graph::initialize, with the P ranks played
one after the other in a single process (same C-ABI calls and re-indexing as rala_amd/multi.py,
the collectives replaced by list shuffling), against the oracle."""
import numpy as np
import pytest
import torch

from rala_amd import hip, multi
from rala_amd.synth import Dataset

import parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("n,g,seed", [(1000, 200_000, 1), (2000, 1_200_000, 33)])
def test_sharded_initialize_matches_oracle(hip_ctx_factory, world, n, g, seed):
    ds = Dataset(n, g, seed)
    st = parity.oracle_stages(ds)
    ov = ds.overlaps
    cuts = multi.slice_starts(ov.a_id, world)
    dev = torch.device("cuda", 0)
    sent = []           # sent[k] = (local_read, bound, counts) of rank k
    valid_parts = []
    for k in range(world):
        sl = ov.take(slice(cuts[k], cuts[k + 1]))
        assert cuts[k] == 0 or ov.a_id[cuts[k]] != ov.a_id[cuts[k] - 1]
        cs = hip_ctx_factory()
        cs.set_reads(ds.read_len)
        cs.set_overlaps(sl)
        cs.dedupe()
        valid_parts.append(cs.valid())
        t_r = torch.empty(4 * max(1, len(sl)), dtype=torch.int32, device=dev)
        t_b = torch.empty_like(t_r)
        cs.emit_bound_tuples(t_r.data_ptr(), t_b.data_ptr())
        reads = t_r[: 4 * len(sl)].to(torch.int64) & 0xFFFFFFFF
        bounds = t_b[: 4 * len(sl)].to(torch.int64) & 0xFFFFFFFF
        sent.append(multi.owner_split(reads, bounds, world))
    parts = {key: [] for key in ("begin", "end", "median", "p10", "alive")}
    pc, pf, hc, hf = [], [], [], []
    for k in range(world):
        # what the all-to-all delivers to owner k
        lr, bd = [], []
        for src in range(world):
            r, b, c = sent[src]
            off = int(c[:k].sum())
            lr.append(r[off: off + int(c[k])])
            bd.append(b[off: off + int(c[k])])
        lr = torch.cat(lr).to(torch.int32)
        bd = torch.cat(bd).to(torch.int32)
        cl = hip_ctx_factory()
        cl.set_reads(np.ascontiguousarray(ds.read_len[k::world]))
        cl.set_bound_tuples_device(lr.data_ptr(), bd.data_ptr(), lr.numel())
        torch.cuda.synchronize()
        cl.initialize()
        p = cl.piles()
        for key in parts:
            parts[key].append(p[key])
        pits, hills = cl.intervals(0), cl.intervals(1)
        pc.append(np.diff(pits[0].astype(np.int64)))
        pf.append(np.concatenate([pits[1].astype(np.int64), pits[2].astype(np.int64)[:, None]], axis=1).reshape(-1))
        hc.append(np.diff(hills[0].astype(np.int64)))
        hf.append(hills[1].astype(np.int64).reshape(-1))
        # coverage of a few owned reads against the oracle
        for r, want in list(st["data0"].items())[:40]:
            if r % world == k:
                parity.assert_same("pile_data[%d]" % r, cl.pile_data(r // world), want)
    piles = {key: multi.interleave(parts[key], n, world) for key in parts}
    p_off, p_flat = multi.merge_intervals(pc, pf, n, world, 3)
    h_off, h_flat = multi.merge_intervals(hc, hf, n, world, 2)
    valid = np.concatenate(valid_parts)
    parity.assert_same("valid", valid, st["valid"])
    for key in parts:
        parity.assert_same("piles0." + key, piles[key], st["piles0"][key])
    parity.assert_same("pits0.offsets", p_off, st["pits0"][0])
    parity.assert_same("pits0.pairs", p_flat[:, :2], st["pits0"][1])
    parity.assert_same("hills0.offsets", h_off, st["hills0"][0])
    parity.assert_same("hills0.pairs", h_flat, st["hills0"][1])
    cg = hip_ctx_factory()
    cg.set_reads(ds.read_len)
    cg.set_overlaps(ov)
    cg.import_state(valid, piles, (p_off, p_flat[:, :2].astype(np.uint32), p_flat[:, 2].astype(np.uint32)),
                    (h_off, h_flat.astype(np.uint32), None))
    cg.construct()
    parity.check_construct(cg, st)
    parity.check_tr(cg, st)


def simulate_sharded(hip_ctx_factory, ds, world, check_buckets=False):
    """The path bench.py runs for WORLD_SIZE > 1 (ShardedRunner.step) with the ranks played one
    after the other on one GPU: owner-grouped tuples straight from the kernel, packed per-read
    state and interval pools "gathered" device to device, rala_hip_import_state_device.  Returns
    the context that holds all reads and overlaps, state imported."""
    ov = ds.overlaps
    n = ds.n_reads
    cuts = multi.slice_starts(ov.a_id, world)
    dev = torch.device("cuda", 0)
    sent, valid_parts = [], []
    for k in range(world):
        sl = ov.take(slice(cuts[k], cuts[k + 1]))
        cs = hip_ctx_factory()
        cs.set_reads(ds.read_len)
        cs.set_overlaps(sl)
        cs.dedupe()
        t_r = torch.empty(4 * max(1, len(sl)), dtype=torch.int32, device=dev)
        t_b = torch.empty_like(t_r)
        counts = cs.emit_bound_tuples_bucketed(world, t_r.data_ptr(), t_b.data_ptr())
        if check_buckets:
            # the buckets hold exactly the tuples of the unbucketed emission
            u_r, u_b = torch.empty_like(t_r), torch.empty_like(t_b)
            cs.emit_bound_tuples(u_r.data_ptr(), u_b.data_ptr())
            ur = u_r[: 4 * len(sl)].cpu().numpy().view(np.uint32).astype(np.int64)
            ub = u_b[: 4 * len(sl)].cpu().numpy().view(np.uint32).astype(np.int64)
            keep = ur != multi.NO_READ
            off = 0
            for p in range(world):
                c = int(counts[p])
                got = sorted(zip(t_r[off: off + c].cpu().numpy().view(np.uint32).tolist(),
                                 t_b[off: off + c].cpu().numpy().view(np.uint32).tolist()))
                m = keep & (ur % world == p)
                assert got == sorted(zip((ur[m] // world).tolist(), ub[m].tolist()))
                off += c
        sent.append((t_r, t_b, counts))
        v = torch.empty(max(1, len(sl)), dtype=torch.uint8, device=dev)
        if len(sl):
            cs.copy_device_state(valid=v.data_ptr())
        valid_parts.append(v[: len(sl)])
    nl = multi.padded_local(n, world)
    off_f, total = multi.state_layout(nl)
    rows = torch.zeros((world, total), dtype=torch.uint8, device=dev)
    pools = []
    for k in range(world):
        lr, bd = [], []
        for src in range(world):
            r, b, c = sent[src]
            o = int(c[:k].sum())
            lr.append(r[o: o + int(c[k])])
            bd.append(b[o: o + int(c[k])])
        lr, bd = torch.cat(lr).contiguous(), torch.cat(bd).contiguous()
        cl = hip_ctx_factory()
        cl.set_reads(np.ascontiguousarray(ds.read_len[k::world]))
        torch.cuda.synchronize()
        cl.set_bound_tuples_device(lr.data_ptr(), bd.data_ptr(), lr.numel())
        cl.initialize()
        n_pool = int(cl.device_state().pool_count)
        pool = torch.empty(max(1, n_pool * multi.POOL_RECORD), dtype=torch.uint8, device=dev)
        cl.copy_device_state(pool=pool.data_ptr(), pool_count=n_pool,
                             **{f: rows[k].data_ptr() + o for f, o in off_f.items()})
        pools.append(pool[: n_pool * multi.POOL_RECORD])
        cl.close()
    counts = [p.numel() // multi.POOL_RECORD for p in pools]
    state = multi.unpack_state(rows, nl, n, counts)
    pool_all = torch.cat(pools)
    valid = torch.cat(valid_parts)
    torch.cuda.synchronize()
    cg = hip_ctx_factory()
    cg.set_reads(ds.read_len)
    cg.set_overlaps(ov)
    cg.import_state_device(pool=pool_all.data_ptr(), pool_count=sum(counts), valid=valid.data_ptr(),
                           **{f: t.data_ptr() for f, t in state.items()})
    return cg


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("n,g,seed", [(1000, 200_000, 1), (2000, 1_200_000, 33),
                                      (1500, 12_000, 3)])      # ~750x: reads beyond a fixed event slot
def test_sharded_device_path_matches_oracle(hip_ctx_factory, world, n, g, seed):
    ds = Dataset(n, g, seed)
    st = parity.oracle_stages(ds)
    cg = simulate_sharded(hip_ctx_factory, ds, world, check_buckets=True)
    parity.assert_same("valid", cg.valid(), st["valid"])
    p = cg.piles()
    for key in ("begin", "end", "median", "p10", "alive"):
        parity.assert_same("piles0." + key, p[key], st["piles0"][key])
    pits, hills = cg.intervals(0), cg.intervals(1)
    parity.assert_same("pits0.offsets", pits[0], st["pits0"][0])
    parity.assert_same("pits0.pairs", pits[1], st["pits0"][1])
    parity.assert_same("hills0.offsets", hills[0], st["hills0"][0])
    parity.assert_same("hills0.pairs", hills[1], st["hills0"][1])
    cg.construct()
    parity.check_construct(cg, st)
    parity.check_tr(cg, st)


def test_sharded_runner_through_rccl_world1(tmp_path):
    """bench.py's WORLD_SIZE > 1 runner (rala_amd/multi.py ShardedRunner: RCCL all-to-all and
    all-gathers, device-to-device state import) launched through torch.distributed.run with one
    rank: same transitive-pair count as the single-context path."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RALA_FORCE_SHARDED="1")
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "c2", "--steps", "1",
            "--warmup", "1", "--no-cpu-baseline"]
    single = subprocess.run(base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root)
    assert single.returncode == 0, single.stderr.decode()[-2000:]
    want = json.loads(single.stdout.decode().strip().splitlines()[-1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29547"] + base[1:]
    sharded = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root, env=env)
    assert sharded.returncode == 0, sharded.stderr.decode()[-2000:]
    got = json.loads(sharded.stdout.decode().strip().splitlines()[-1])
    assert got["config"]["transitive_pairs"] == want["config"]["transitive_pairs"] > 0
    assert "exchange_ms" in got["stage_ms"] and "gather_ms" in got["stage_ms"]
