"""Python side of the sharded run: one process per GPU (bench.py under torchrun, ShardedRunner) or
all ranks as host threads of this process (bare `bench.py --gpus N`, ThreadedRunner - what
`rala --gpus N` does in C++, rala_amd/host/graph.cpp).

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
import threading

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
        self.cuts = hip.slice_cuts(ov.a_id, world, ov.b_id)
        lo, hi = my_slice(self.cuts, rank)
        err = None
        self.mg = None
        # every rank's device contexts first (not collective); only when all ranks have theirs does anybody enter
        # ncclCommInitRank - a rank that failed before it would leave the others waiting inside
        try:
            self.mg = hip.ShardedRank(local_rank, rank, world)
        except Exception as e:          # noqa: BLE001 - agreed on below, then raised
            err = e
        if not all_agree(err is None, group):
            if self.mg is not None:
                self.mg.close()
            raise err if err is not None else RuntimeError("another rank failed to set up its GPU")
        try:
            uid = exchange_id(hip.unique_id, group)
            self.mg.join(uid)                                            # collective: ncclCommInitRank
            self.mg.set_reads(ds.read_len)
            self._slice = ov.take(slice(lo, hi))
            self.mg.set_overlaps(self._slice, lo)
        except Exception as e:          # noqa: BLE001 - agreed on below, then raised
            err = e
        if not all_agree(err is None, group):
            raise err if err is not None else RuntimeError("another rank failed to join the group")
        self._tm = {}

    def step(self):
        n_tr = self.mg.run()
        self._tm = rank_timings(self.mg)
        return n_tr

    def timings(self):
        return self._tm

    def close(self):
        self.mg.close()


_STAGE_KEYS = ("dedupe_ms", "bucket_ms", "pile_ms", "pile_launches", "pile_overflow_reads", "pile_position_reads")
_CONSTRUCT_KEYS = ("classify_ms", "death_ms", "finish_ms", "tail_host_ms", "death_rounds")


def rank_timings(mg):
    """one rank's stage timings in the names bench.py's single-GPU line uses"""
    tm = dict(mg.timings())
    tm.update({"owner_" + k: v for k, v in mg.owner_timings().items() if k in _STAGE_KEYS})
    tm.update({k: v for k, v in mg.context().timings().items() if k in _CONSTRUCT_KEYS})
    tm["pile_ms"] = tm.get("owner_pile_ms", 0.0)
    tm["bucket_ms"] = tm.get("owner_bucket_ms", 0.0)
    tm["dedupe_ms"] = 0.0
    return tm


class ThreadedRunner:
    """All ranks in THIS process, one host thread per rank / GPU (no launcher needed).

    transport "rccl": every rank joins one RCCL communicator (ncclCommInitRank is collective, so
    the rank objects are created on threads of their own; ctypes releases the GIL inside the
    call).  transport "local": the in-process transport (peer copies between the devices; also
    several ranks on one device, devices = [0, 0, ...]).  A step is rala_hip_mg_run_threads."""

    def __init__(self, ds, world, devices=None, transport="rccl", join_timeout=None):
        import os

        from . import hip

        self.hip = hip
        self.world = world
        self.transport = transport
        devices = list(devices) if devices is not None else list(range(world))
        assert len(devices) == world
        ov = ds.overlaps
        self.cuts = hip.slice_cuts(ov.a_id, world, ov.b_id)
        self._group = None
        if transport == "local":
            self._group = hip.LocalGroup(world)
            token = self._group
        elif transport == "rccl":
            token = hip.unique_id()
        else:
            raise ValueError("transport is 'rccl' or 'local'")
        self.ranks = [None] * world
        self._slices = [None] * world
        errs = [None] * world

        def contexts(k):
            try:
                self.ranks[k] = hip.ShardedRank(devices[k], k, world)
            except Exception as e:      # noqa: BLE001 - reported below, for all ranks together
                errs[k] = e

        def join(k):
            try:
                mg = self.ranks[k]
                mg.join(token)
                mg.set_reads(ds.read_len)
                lo, hi = my_slice(self.cuts, k)
                self._slices[k] = ov.take(slice(lo, hi))
                mg.set_overlaps(self._slices[k], lo)
            except Exception as e:      # noqa: BLE001
                errs[k] = e

        def check(what):
            bad = [(k, e) for k, e in enumerate(errs) if e is not None]
            if bad:
                self.close()
                raise RuntimeError("sharded set-up failed (%s): " % what + "; ".join("rank %d: %s" % (k, e) for k, e in bad))

        # 1. every rank's device contexts (not collective): nobody enters ncclCommInitRank unless all ranks have theirs
        th = [threading.Thread(target=contexts, args=(k,)) for k in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        check("device contexts")
        # 2. joining is collective; a join that does not come back (RCCL cannot be interrupted from outside) is reported
        # after a time limit instead of waited for - the stuck threads are daemons and end with the process
        limit = float(join_timeout if join_timeout is not None else os.environ.get("RALA_JOIN_TIMEOUT", "300"))
        th = [threading.Thread(target=join, args=(k,), daemon=True) for k in range(world)]
        for t in th:
            t.start()
        import time
        deadline = time.monotonic() + limit
        for t in th:
            t.join(max(0.0, deadline - time.monotonic()))
        stuck = [k for k, t in enumerate(th) if t.is_alive()]
        if stuck:
            self.ranks = []             # (their objects are in use by the stuck threads: not destroyed)
            raise TimeoutError("sharded set-up: ranks %s did not join the %s group within %.0f s" % (stuck, transport, limit))
        check("joining the group")
        self._tm = {}

    def step(self, sens_slices=None):
        n_tr = self.hip.run_ranks(self.ranks, sens_slices)
        per_rank = [rank_timings(mg) for mg in self.ranks]
        # the slowest rank's figure per stage (the step waits for it)
        self._tm = {k: max(float(t.get(k, 0.0)) for t in per_rank) for k in per_rank[0]}
        return n_tr

    def timings(self):
        return self._tm

    def close(self):
        for mg in self.ranks:
            if mg is not None:
                mg.close()
        self.ranks = []
        if self._group is not None:
            self._group.close()
            self._group = None
