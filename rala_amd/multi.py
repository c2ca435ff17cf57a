"""Multi-GPU sharding of Graph::initialize (one process per GPU, torch.distributed / RCCL).

Partition: read r is owned by rank r % P and is local read r // P there.  The overlap file is
cut into P contiguous slices on a_id-run boundaries (duplicate removal is per run,
reference graph.cpp:346-350).  Per step:

  1. every rank removes duplicates in its slice and turns the slice's overlaps into bound
     tuples (read, bound)                                   [rala_hip_dedupe / _emit_bound_tuples]
  2. ONE all-to-all(v) ships every tuple to the owner of its read (coverage is additive
     mod 2^16, so arrival order is irrelevant)
  3. owners bucket the tuples and build + annotate their piles   [rala_hip_set_bound_tuples,
     rala_hip_initialize]
  4. all-gather of the per-read annotations (16 B per read + the few pits / hills) and of the
     validity bits; every rank installs them                     [rala_hip_import_state]
  5. the remainder (second overlap pass, containment fixed point, preprocess tail, graph,
     transitive reduction) is small and runs replicated on every rank.

The tensor plumbing below (owner split, variable all-to-all, padded all-gather, global
re-indexing) is device agnostic so that the CPU test-suite can run it under gloo.
"""
import numpy as np
import torch
import torch.distributed as dist

NO_READ = 0xFFFFFFFF


def slice_starts(a_id, world):
    """Cut points of the overlap arrays: world + 1 indices, every cut on an a_id change."""
    n = len(a_id)
    cuts = [0]
    for k in range(1, world):
        i = max(cuts[-1], (n * k) // world)
        while 0 < i < n and a_id[i] == a_id[i - 1]:
            i += 1
        cuts.append(min(i, n))
    cuts.append(n)
    return cuts


def n_local_reads(n_reads, rank, world):
    return (n_reads - rank + world - 1) // world if n_reads > rank else 0


def owner_split(reads, bounds, world):
    """reads / bounds: int64 tensors of equal length (reads == NO_READ dropped).
    Returns (local_read, bound, counts_per_owner) ordered by owner."""
    keep = reads != NO_READ
    reads, bounds = reads[keep], bounds[keep]
    owner = reads % world
    order = torch.argsort(owner, stable=True)
    counts = torch.bincount(owner, minlength=world)
    return (reads[order] // world), bounds[order], counts


def all_to_all_v(x, send_counts, group=None):
    """Variable all-to-all of a 1-D tensor; returns (received, recv_counts)."""
    world = dist.get_world_size(group)
    send_counts = send_counts.to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    s, r = send_counts.tolist(), recv_counts.tolist()
    out = torch.empty(int(sum(r)), dtype=x.dtype, device=x.device)
    dist.all_to_all_single(out, x.contiguous(), output_split_sizes=r, input_split_sizes=s, group=group)
    assert len(s) == world
    return out, recv_counts


def all_gather_v(x, group=None):
    """All-gather of 1-D tensors of different lengths; returns the list of per-rank tensors."""
    world = dist.get_world_size(group)
    n = torch.tensor([x.numel()], dtype=torch.int64, device=x.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(t.item()) for t in ns]
    m = max(ns) if ns else 0
    pad = torch.zeros(max(m, 1), dtype=x.dtype, device=x.device)
    pad[: x.numel()] = x
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return [o[:k] for o, k in zip(outs, ns)]


def interleave(parts, n_reads, world):
    """parts[k][j] belongs to global read j * world + k."""
    out = np.zeros(n_reads, dtype=parts[0].dtype)
    for k in range(world):
        out[k::world] = parts[k][: n_local_reads(n_reads, k, world)]
    return out


def merge_intervals(counts_parts, flat_parts, n_reads, world, width):
    """Per-rank interval CSR (counts per local read, flat rows of `width` columns) ->
    global (offsets[n_reads + 1], flat) in global read order."""
    counts = interleave(counts_parts, n_reads, world).astype(np.uint64)
    offs = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(counts, out=offs[1:])
    owners, rows = [], []
    for k in range(world):
        c = counts_parts[k][: n_local_reads(n_reads, k, world)].astype(np.int64)
        local = np.repeat(np.arange(len(c), dtype=np.int64), c)
        owners.append(local * world + k)
        rows.append(np.asarray(flat_parts[k]).reshape(-1, width))
    if owners:
        g = np.concatenate(owners)
        flat = np.concatenate(rows) if rows else np.zeros((0, width), dtype=np.uint32)
        order = np.argsort(g, kind="stable")
        flat = flat[order]
    else:
        flat = np.zeros((0, width), dtype=np.uint32)
    return offs, flat


class ShardedRunner:
    """bench.py's runner for WORLD_SIZE > 1 (one process per GPU)."""

    def __init__(self, ds, rank, world, local_rank, group=None):
        from . import hip
        from .synth import Overlaps

        self.hip = hip
        self.rank, self.world, self.group = rank, world, group
        self.n_reads = ds.n_reads
        self.dev = torch.device("cuda", local_rank)
        ov = ds.overlaps
        cuts = slice_starts(ov.a_id, world)
        lo, hi = cuts[rank], cuts[rank + 1]
        self.slice = ov.take(slice(lo, hi))
        self.slice_lens = [cuts[k + 1] - cuts[k] for k in range(world)]
        # slice context: duplicate removal + tuple emission for this rank's overlaps
        self.cs = hip.Context(local_rank)
        self.cs.set_reads(ds.read_len)
        self.cs.set_overlaps(self.slice)
        # owner context: the reads this rank owns
        self.local_len = np.ascontiguousarray(ds.read_len[rank::world])
        self.cl = hip.Context(local_rank)
        self.cl.set_reads(self.local_len)
        # replicated context for everything after initialize
        self.cg = hip.Context(local_rank)
        self.cg.set_reads(ds.read_len)
        self.cg.set_overlaps(ov)
        n4 = 4 * max(1, len(self.slice))
        self.t_reads = torch.empty(n4, dtype=torch.int32, device=self.dev)
        self.t_bounds = torch.empty(n4, dtype=torch.int32, device=self.dev)
        self._tm = {}

    def step(self):
        hip, world = self.hip, self.world
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        # 1. duplicates + tuples of this slice
        self.cs.dedupe()
        valid_slice = torch.from_numpy(self.cs.valid()).to(self.dev)
        self.cs.emit_bound_tuples(self.t_reads.data_ptr(), self.t_bounds.data_ptr())
        n4 = 4 * len(self.slice)
        reads = self.t_reads[:n4].to(torch.int64) & 0xFFFFFFFF
        bounds = self.t_bounds[:n4].to(torch.int64) & 0xFFFFFFFF
        ev[0].record()
        # 2. one all-to-all to the owners
        lr, bd, counts = owner_split(reads, bounds, world)
        lr, _ = all_to_all_v(lr.to(torch.int32), counts, self.group)
        bd, _ = all_to_all_v(bd.to(torch.int32), counts, self.group)     # bit pattern of the uint32 bound
        ev[1].record()
        # 3. owners build their piles
        self._keep = (lr, bd)
        torch.cuda.synchronize()
        self.cl.set_bound_tuples_device(lr.data_ptr(), bd.data_ptr(), lr.numel())
        try:
            self.cl.initialize()
        except hip.RalaHipError as e:
            if e.code != -4:        # every local read filtered is not fatal for the whole job
                raise
        p = self.cl.piles()
        pits = self.cl.intervals(0)
        hills = self.cl.intervals(1)
        ev[2].record()
        # 4. all-gather annotations + validity bits, install them
        def gather_np(a, dtype):
            t = torch.from_numpy(np.ascontiguousarray(a).astype(dtype, copy=False)).to(self.dev)
            return [x.cpu().numpy() for x in all_gather_v(t, self.group)]

        n = self.n_reads
        piles = {k: interleave(gather_np(p[k].astype(np.int64), np.int64), n, world).astype(p[k].dtype)
                 for k in ("begin", "end", "median", "p10", "alive")}
        pc = gather_np(np.diff(pits[0].astype(np.int64)), np.int64)
        pf = gather_np(np.concatenate([pits[1].astype(np.int64), pits[2].astype(np.int64)[:, None]],
                                      axis=1).reshape(-1), np.int64)
        hc = gather_np(np.diff(hills[0].astype(np.int64)), np.int64)
        hf = gather_np(hills[1].astype(np.int64).reshape(-1), np.int64)
        p_off, p_flat = merge_intervals(pc, pf, n, world, 3)
        h_off, h_flat = merge_intervals(hc, hf, n, world, 2)
        valid = np.concatenate([x.cpu().numpy() for x in all_gather_v(valid_slice, self.group)])
        self.cg.import_state(valid, piles,
                             (p_off, p_flat[:, :2].astype(np.uint32), p_flat[:, 2].astype(np.uint32)),
                             (h_off, h_flat.astype(np.uint32), None))
        ev[3].record()
        # 5. replicated remainder
        self.cg.construct()
        n_tr = self.cg.remove_transitive_edges()
        torch.cuda.synchronize()
        tl, tg = self.cl.timings(), self.cg.timings()
        self._tm = dict(tg)
        for k in ("dedupe_ms", "bucket_ms", "pile_ms", "pile_launches", "pile_overflow_reads", "pile_position_reads"):
            self._tm[k] = tl[k]
        self._tm["exchange_ms"] = ev[0].elapsed_time(ev[1])
        self._tm["gather_ms"] = ev[2].elapsed_time(ev[3])
        return n_tr

    def timings(self):
        return self._tm
