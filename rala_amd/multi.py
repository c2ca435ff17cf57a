"""One process per GPU (bench.py under torchrun): Python side of the sharded run.

The data path is C++ (rala_amd/csrc/sharded.hip: rala_hip_mg_*, RCCL called directly on the
context's stream).  What is left for Python when every rank is its own process:

  * rendezvous: rank 0 obtains the 128-byte RCCL id and ships it to the others; torch.distributed
    (gloo, CPU) is the channel - it also carries bench.py's barrier and its max-over-ranks clock;
  * every rank cuts the same slices out of the overlap file (rala_hip_mg_slice_cuts) and hands
    its own to its rank object.

The same two steps run under gloo with world size 2 in the CPU test-suite
(tests/test_multi_cpu.py); the collectives themselves need GPUs and are covered by the rank
simulation of tests/test_gpu_sharded.py.
"""
import numpy as np
import torch
import torch.distributed as dist


def broadcast_bytes(payload, src=0, group=None):
    """the same bytes object on every rank (rank `src` supplies it, the others pass None)"""
    box = [payload if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def exchange_id(make_id, group=None):
    """rank 0 calls make_id() (-> 128 bytes), every rank returns the same bytes"""
    rank = dist.get_rank(group)
    payload = make_id() if rank == 0 else None
    if rank == 0 and (not isinstance(payload, (bytes, bytearray)) or len(payload) != 128):
        raise ValueError("the RCCL id is 128 bytes")
    return bytes(broadcast_bytes(payload, 0, group))


def my_slice(cuts, rank):
    return int(cuts[rank]), int(cuts[rank + 1])


def all_agree(ok, group=None):
    """True on every rank iff every rank passed True (a rank that failed to set up must not leave
    the others waiting in the first collective)"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


def max_over_ranks(x, group=None):
    t = torch.tensor([float(x)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class ShardedRunner:
    """bench.py's runner for WORLD_SIZE > 1: one rala_hip_mg rank object per process"""

    def __init__(self, ds, rank, world, local_rank, group=None):
        from . import hip

        self.hip = hip
        self.rank, self.world = rank, world
        ov = ds.overlaps
        self.cuts = hip.slice_cuts(ov.a_id, world)
        lo, hi = my_slice(self.cuts, rank)
        err = None
        try:
            uid = exchange_id(hip.unique_id, group)
            self.mg = hip.ShardedRank(local_rank, rank, world, uid)       # collective: ncclCommInitRank
            self.mg.set_reads(ds.read_len)
            self._slice = ov.take(slice(lo, hi))
            self.mg.set_overlaps(self._slice, lo)
        except Exception as e:          # noqa: BLE001 - agreed on below, then raised
            err = e
        if not all_agree(err is None, group):
            raise err if err is not None else RuntimeError("another rank failed to set up its GPU")
        self._tm = {}

    def step(self):
        n_tr = self.mg.run()
        tm = dict(self.mg.timings())
        tm.update({"owner_" + k: v for k, v in self.mg.owner_timings().items()
                   if k in ("dedupe_ms", "bucket_ms", "pile_ms", "pile_launches", "pile_overflow_reads", "pile_position_reads")})
        tm.update({k: v for k, v in self.mg.context().timings().items()
                   if k in ("classify_ms", "death_ms", "finish_ms", "tail_host_ms", "death_rounds")})
        tm["pile_ms"] = tm.get("owner_pile_ms", 0.0)
        tm["bucket_ms"] = tm.get("owner_bucket_ms", 0.0)
        tm["dedupe_ms"] = 0.0
        self._tm = tm
        return n_tr

    def timings(self):
        return self._tm

    def close(self):
        self.mg.close()
