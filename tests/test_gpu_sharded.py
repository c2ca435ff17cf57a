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
