"""Multi-GPU sharding of Graph::initialize (one process per GPU, torch.distributed / RCCL).

Partition: read r is owned by rank r % P and is local read r // P there.  The overlap file is
cut into P contiguous slices on a_id-run boundaries (duplicate removal is per run,
reference graph.cpp:346-350).  Per step:

  1. every rank removes duplicates in its slice and turns the slice's overlaps into bound
     tuples (local read, bound) grouped by owner  [rala_hip_dedupe, _emit_bound_tuples_bucketed]
  2. ONE all-to-all(v) ships every tuple to the owner of its read (coverage is additive
     mod 2^16, so arrival order is irrelevant)
  3. owners bucket the tuples and build + annotate their piles   [rala_hip_set_bound_tuples,
     rala_hip_initialize]
  4. all-gather of the per-read annotations (19 B per read, one packed buffer per rank), of the
     interval pools and of the validity bits, all device to device; every rank installs the
     result                                        [rala_hip_copy_device_state, _import_state_device]
  5. the remainder (second overlap pass, containment fixed point, preprocess tail, graph,
     transitive reduction) is small and runs replicated on every rank.

The tensor plumbing below (variable all-to-all, padded all-gather, packed state layout, global
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


# packed per-read state of one rank: field arrays of nl entries back to back (nl % 8 == 0)
STATE_FIELDS = (("begin", 4), ("end", 4), ("slot", 4), ("median", 2), ("p10", 2), ("alive", 1), ("n_pits", 1),
                ("n_hills", 1))
_VIEW = {4: torch.int32, 2: torch.int16, 1: torch.uint8}
NO_SLOT = -1        # 0xFFFFFFFF seen as int32
POOL_RECORD = 12    # bytes of one {first, second, aux} interval


def padded_local(n_reads, world):
    """entries per rank in the packed state (largest local read count rounded up to 8)"""
    return (n_local_reads(n_reads, 0, world) + 7) // 8 * 8


def state_layout(nl):
    """byte offset of every field in the packed state, and its total size"""
    off, o = {}, 0
    for name, w in STATE_FIELDS:
        off[name] = o
        o += w * nl
    return off, o


def all_gather_rows(x, group=None):
    """x: 1-D tensor of the same length on every rank -> (world, len) tensor"""
    world = dist.get_world_size(group)
    out = torch.empty((world, x.numel()), dtype=x.dtype, device=x.device)
    dist.all_gather(list(out.unbind(0)), x.contiguous(), group=group)
    return out


def unpack_state(rows, nl, n_reads, pool_counts):
    """rows: (world, bytes) uint8, rank k's packed state in row k.  Returns the global per-read
    tensors (read j * world + k = entry j of rank k); slots are rebased onto the concatenation
    of the ranks' interval pools (pool_counts = records per rank)."""
    world = rows.shape[0]
    off, _ = state_layout(nl)
    base = torch.zeros(world, dtype=torch.int64)
    base[1:] = torch.cumsum(torch.as_tensor(pool_counts, dtype=torch.int64), 0)[:-1]
    base = base.to(device=rows.device, dtype=torch.int32).view(-1, 1)
    out = {}
    for name, w in STATE_FIELDS:
        f = rows[:, off[name]: off[name] + w * nl].view(_VIEW[w])            # (world, nl)
        if name == "slot":
            f = torch.where(f == NO_SLOT, f, f + base)
        out[name] = f.t().reshape(-1)[:n_reads].contiguous()
    return out


def gather_pools(pool, group=None):
    """pool: uint8 tensor of this rank's interval records -> (concatenated pools, records per rank)"""
    parts = all_gather_v(pool, group)
    counts = [p.numel() // POOL_RECORD for p in parts]
    return (torch.cat(parts) if parts else pool), counts


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
        hip, world, dev = self.hip, self.world, self.dev
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        u8 = lambda n: torch.empty(max(int(n), 1), dtype=torch.uint8, device=dev)
        import time
        wall = [time.perf_counter()]

        def lap():
            torch.cuda.synchronize()
            wall.append(time.perf_counter())
        # 1. duplicates + owner-grouped tuples of this slice
        self.cs.dedupe()
        counts = self.cs.emit_bound_tuples_bucketed(world, self.t_reads.data_ptr(), self.t_bounds.data_ptr())
        ev[0].record()
        lap()
        # 2. one all-to-all to the owners
        n_send = int(counts.sum())
        send = torch.from_numpy(counts.astype(np.int64)).to(dev)
        lr, _ = all_to_all_v(self.t_reads[:n_send], send, self.group)
        bd, _ = all_to_all_v(self.t_bounds[:n_send], send, self.group)    # bit pattern of the uint32 bound
        ev[1].record()
        lap()
        # 3. owners build their piles
        self.cl.set_bound_tuples_device(lr.data_ptr(), bd.data_ptr(), lr.numel())
        try:
            self.cl.initialize()
        except hip.RalaHipError as e:
            if e.code != -4:        # every local read filtered is not fatal for the whole job
                raise
        ev[2].record()
        lap()
        # 4. all-gather annotations + interval pools + validity bits (device to device), install them
        n, nl = self.n_reads, padded_local(self.n_reads, world)
        off, total = state_layout(nl)
        packed = u8(total)
        n_pool = int(self.cl.device_state().pool_count)
        pool = u8(n_pool * POOL_RECORD)
        self.cl.copy_device_state(pool=pool.data_ptr(), pool_count=n_pool,
                                  **{k: packed.data_ptr() + o for k, o in off.items()})
        vmax = max(self.slice_lens)
        vbuf = u8(vmax)
        if len(self.slice):
            self.cs.copy_device_state(valid=vbuf.data_ptr())
        rows = all_gather_rows(packed, self.group)
        pools, pool_counts = gather_pools(pool[: n_pool * POOL_RECORD], self.group)
        vrows = all_gather_rows(vbuf, self.group)
        valid = torch.cat([vrows[k, : self.slice_lens[k]] for k in range(world)]) if sum(self.slice_lens) else vbuf
        st = unpack_state(rows, nl, n, pool_counts)
        torch.cuda.synchronize()
        self.cg.import_state_device(pool=pools.data_ptr(), pool_count=sum(pool_counts), valid=valid.data_ptr(),
                                    **{k: t.data_ptr() for k, t in st.items()})
        ev[3].record()
        lap()
        # 5. replicated remainder
        self.cg.construct()
        n_tr = self.cg.remove_transitive_edges()
        lap()
        tl, tg = self.cl.timings(), self.cg.timings()
        self._tm = dict(tg)
        for k in ("dedupe_ms", "bucket_ms", "pile_ms", "pile_launches", "pile_overflow_reads", "pile_position_reads"):
            self._tm[k] = tl[k]
        for k, name in enumerate(("emit", "exchange", "owner_init", "gather", "remainder")):
            self._tm["wall_%s_ms" % name] = 1e3 * (wall[k + 1] - wall[k])
        self._tm["exchange_ms"] = ev[0].elapsed_time(ev[1])
        self._tm["gather_ms"] = ev[2].elapsed_time(ev[3])
        return n_tr

    def timings(self):
        return self._tm
