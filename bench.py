#!/usr/bin/env python3
"""Throughput of Rala's hot path on MI355X: overlaps/sec for pile build + filtering +
graph construction + transitive reduction (BASELINE.json metric), synthetic input of the
named shape resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c1|c5|c2s|c3s|c5s]

One step = one pass of the whole hot path over the data set: rala_hip_initialize
(duplicate removal, bound bucketing, pile build + annotation), rala_hip_construct
(second overlap pass, containment fixed point, preprocess tail, graph build) and
rala_hip_remove_transitive_edges.  Prints ONE JSON line on rank 0.  The workloads c3s / c5s add
the sensitive second pass (`rala -s`, reference graph.cpp:882-1054) to every step: the sensitive
overlaps - derived from the piles of an untimed first pass, as the two-run workflow of the
reference derives them from its `-p` output - lie in HBM like the primary ones.

Several GPUs (reads hash-partitioned, ONE all-to-all of bound tuples; rala_hip_mg_*):
  * bare `python bench.py --gpus N`: the N ranks are host threads of this process, one per
    GPU, over RCCL - what `rala --gpus N` does (rala_amd/host/graph.cpp);
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: one
    process per GPU; torch carries only the 128-byte RCCL id, the barrier and the clock (gloo).
The line says which transport carried the ranks ("transport", "rccl_ranks").  A run that was asked
for RCCL and did not get it ends with a non-zero exit code and no line; `--transport local` (peer
copies inside one process, several ranks may share a device) is a test of the decomposition that
has to be asked for by name.
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A process has four hardware queues by default and its streams share them; a rank of a sharded run has two contexts with
# three streams each beside RCCL's, and kernels meant to run beside each other ended up behind each other (the owner's small
# pile kernels in front of the big one: 0.1 - 0.3 ms per C3 step, docs/history/gpurun/r5_queues.sh).  The runtime reads this when it
# starts - before anything imports torch.  (The single-GPU step has four streams: it stays on the runtime's default, as profiled.)
def _several_ranks(argv):
    n = 1
    for k, a in enumerate(argv):
        if a == "--gpus" and k + 1 < len(argv):
            n = argv[k + 1]
        elif a.startswith("--gpus="):
            n = a.split("=", 1)[1]
    try:
        n = int(n)
    except ValueError:
        n = 1
    return n > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("RALA_FORCE_SHARDED")


if _several_ranks(sys.argv[1:]):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
ACHIEVABLE_GBS = 6300.0 # what streaming kernels reach of it on this part (MI355X_MICROARCH.md; the builder's own row fills: 5.0 - 6.5 TB/s, profiles/README.md)

WORKLOADS = {
    "c1": "1k reads / 50k overlaps (BASELINE configs[0])",
    "c2": "100k reads / 5M overlaps (BASELINE configs[1])",
    "c3": "1M reads / 50M overlaps (BASELINE configs[2])",
    "c5": "4M reads / 300M overlaps (BASELINE configs[4], primary overlaps only)",
    "c2s": "100k reads / 5M overlaps + the sensitive second pass, -s (BASELINE configs[1] with configs[4]'s second pass)",
    "c3s": "1M reads / 50M overlaps + the sensitive second pass, -s (BASELINE configs[2] with configs[4]'s second pass)",
    "c5s": "4M reads / 300M overlaps + the sensitive second pass, -s (BASELINE configs[4])",
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(headline="c3", single="c2"):
    """The CPU path, timed on this box's host cores: as many threads as the process may really use on the
    HEADLINE configuration itself (`headline`: the data set `value` is quoted on, or C3 for the C5 workloads), and one
    thread on a bounded sample (`single`).  One task per pile through a thread pool with the interface of the
    reference's vendor/thread_pool (graph.cpp:367-377, 387-407; oracle Driver::task_pool).  kind "reference": the
    reference's own rala::Pile / rala::Overlap objects (oracle/_ref, compiled from the reference's pile.cpp /
    overlap.cpp) under the restated Graph orchestration; kind "port": the flat restatement (oracle/_build) when that
    library is not there.  Both are test infrastructure used here only as the measured baseline."""
    from oracle import oracle as ora
    from rala_amd.synth import Dataset

    from rala_amd.cpus import effective_cpus

    cores = effective_cpus()        # affinity mask cut by the container's CPU quota (16 on the gpurun boxes)

    def once(ds, ref, threads):
        t0 = time.perf_counter()
        o = ora.Oracle(ds.read_len, ds.overlaps, n_threads=threads, ref=ref, task_pool=True)
        rc = o.construct()
        n_tr = o.remove_transitive_edges() if rc == 0 else 0
        dt = time.perf_counter() - t0
        del o
        return dt, n_tr

    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liboracle_ref.so"))
    out = {"unit": "overlaps/s", "cores": cores}
    ds = Dataset.config(headline)
    n = len(ds.overlaps)
    kind = "reference" if have_ref else "port"
    try:
        dt, n_tr = once(ds, have_ref, cores)
    except Exception as e:              # the library is there but unusable on this box: say so, fall back
        if not have_ref:
            raise
        log("[bench] reference-object baseline failed (%s); using the port" % e)
        kind = "port"
        dt, n_tr = once(ds, False, cores)
    n_reads = ds.n_reads
    if single != headline:
        ds = Dataset.config(single)
    n1 = len(ds.overlaps)
    dt1, n_tr1 = once(ds, kind == "reference", 1)
    if single == headline:
        assert n_tr1 == n_tr
    out.update(value=n / dt, kind=kind, value_1_thread=n1 / dt1,
               sample="%s synthetic (the headline configuration itself), %d reads / %d overlaps, whole hot path once with %s, "
                      "one task per pile through a thread pool: %.2f s on %d threads, %d transitive pairs; one thread: %s "
                      "synthetic, %d overlaps in %.2f s" % (
                          headline, n_reads, n, "the reference's Pile / Overlap objects" if kind == "reference"
                          else "the flat restatement", dt, cores, n_tr, single, n1, dt1))
    return out


def end_to_end_from_paf(ds, workload):
    """The second figure of SURVEY 8(d), measured in THIS run: the data set written as PAF text to a scratch file, then from
    that text to the reduced graph (rala_e2e_from_paf_with in librala.so; best of three).  Ingest is the device tokeniser -
    the file's text shipped to the device and parsed there, one thread per line (rala_hip_set_overlaps_from_paf) - and, for
    comparison, the host readers (multi-threaded parse, then the columns' upload)."""
    from rala_amd import build
    from rala_amd.cpus import effective_cpus

    build.build_host()
    L = ctypes.CDLL(os.path.join(build.PKG, "host", "librala.so"))
    L.rala_e2e_from_paf_with.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int] + [ctypes.c_void_p] * 6
    threads = effective_cpus()
    read_len = np.ascontiguousarray(ds.read_len, dtype=np.uint32)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        paf = os.path.join(d, "ovl.paf")
        t0 = time.perf_counter()
        ds.write_paf(paf)
        size = os.path.getsize(paf)
        log("[bench] end to end: wrote %.2f GB of PAF in %.1f s" % (size / 1e9, time.perf_counter() - t0))

        def best_of(device_ingest, runs):
            best = None
            for _ in range(runs):
                ms = [ctypes.c_double() for _ in range(3)]
                n_ovl, n_tr, used = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_int()
                rc = L.rala_e2e_from_paf_with(paf.encode(), read_len.ctypes.data, ds.n_reads, threads, device_ingest,
                                              *[ctypes.byref(x) for x in ms], ctypes.byref(n_ovl), ctypes.byref(n_tr), ctypes.byref(used))
                if rc != 0:
                    raise RuntimeError("rala_e2e_from_paf: %d" % rc)
                tot = sum(x.value for x in ms)
                if best is None or tot < best["ms_total"]:
                    best = {"value": n_ovl.value / (tot * 1e-3), "unit": "overlaps/s", "threads": threads, "paf_bytes": size,
                            "ms_parse": ms[0].value, "ms_upload": ms[1].value, "ms_device_first_call": ms[2].value,
                            "ms_total": tot, "transitive_pairs": n_tr.value,
                            "ingest": "device tokeniser" if used.value else "host readers"}
            return best
        best = best_of(1, 4)            # (the first call makes the pinned staging blocks)
        host = best_of(0, 2)
    if best["ingest"] == "device tokeniser":
        best["source"] = ("measured in this run (best of 4): PAF text -> device memory (reader threads, pinned staging: ms_parse holds "
                          "this and the tokeniser) -> tokenised on the device -> device path; ms_upload = the name table")
    else:
        best["source"] = "measured in this run: PAF text -> threaded ingest -> upload -> device path"
    best["host_readers"] = {k: host[k] for k in ("value", "ms_parse", "ms_upload", "ms_device_first_call", "ms_total", "transitive_pairs")}
    assert host["transitive_pairs"] == best["transitive_pairs"]
    return best


def from_pinned_host(ds, device, steps=3):
    """SURVEY 8(d) words the kernel-only figure as "binary SoA already in pinned host memory" - what a Graph that keeps its
    own parser would hand over.  `value` has the columns in HBM; this member is the same step with
    the upload in front of it: the eight columns (29 bytes per overlap) from page-locked host memory over PCIe, then
    initialize + construct + remove_transitive_edges.  The link is the bound (26 ms of transfer against a step of 7.5 at C3);
    what the step can do is hide: every kernel up to the pile chain reads only some of the columns, so with the columns sent in
    the order of their first use (RALA_HIP_MEM_HOST_ASYNC) the bucketing and the pile kernels run while the rest arrives."""
    import torch
    from rala_amd import hip
    from rala_amd.synth import FIELDS

    class _Pinned:
        pass
    pinned = _Pinned()
    keep = []
    for f in list(FIELDS) + ["strand"]:
        t = torch.from_numpy(getattr(ds.overlaps, f)).pin_memory()
        keep.append(t)
        setattr(pinned, f, t.numpy())
    pinned.__class__.__len__ = lambda self: len(ds.overlaps)
    ctx = hip.Context(device)
    ctx.set_reads(ds.read_len)
    n = len(ds.overlaps)
    nbytes = 29.0 * n
    out = {}
    for later in (False, True):
        best = None
        n_tr = 0
        for _ in range(steps + 1):                 # (the first call allocates the columns' device memory)
            t0 = time.perf_counter()
            ctx.set_overlaps(pinned, later=later)
            t1 = time.perf_counter()
            ctx.initialize()
            ctx.construct()
            n_tr = ctx.remove_transitive_edges()
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1)
        if not later:
            out = {"unit": "overlaps/s", "back_to_back": {"value": n / best[0], "ms_upload": 1e3 * best[1], "ms_step": 1e3 * best[2],
                                                          "ms_total": 1e3 * best[0], "upload_GBs": nbytes / best[1] / 1e9},
                   "bytes_uploaded": nbytes, "transitive_pairs": int(n_tr)}
        else:
            assert int(n_tr) == out["transitive_pairs"], (n_tr, out["transitive_pairs"])
            out.update({"value": n / best[0], "ms_total": 1e3 * best[0], "link_GBs": nbytes / best[0] / 1e9})
    ctx.close()
    out["source"] = ("measured in this run (best of %d): columns in page-locked host memory -> RALA_HIP_MEM_HOST_ASYNC: the eight copies "
                     "over PCIe leave inside rala_hip_initialize, each in front of the first kernel that reads its column (ids -> the "
                     "counting pass, b coordinates -> the first scatter, a coordinates -> the query side, lengths and strands beside the "
                     "pile kernels) -> the step; back_to_back = RALA_HIP_MEM_HOST, the upload first, then the step" % steps)
    return out


def end_to_end_from_paf_ranks(ds, world, devices, transport):
    """the same for a sharded run (ranks as threads of this process): every rank ships and tokenises its own byte range of the
    file on its own GPU (rala_hip_mg_set_overlaps_from_paf), then the sharded step - nothing is parsed on the host"""
    from rala_amd import build
    from rala_amd.cpus import effective_cpus

    build.build_host()
    L = ctypes.CDLL(os.path.join(build.PKG, "host", "librala.so"))
    L.rala_e2e_from_paf_ranks.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                          ctypes.c_int] + [ctypes.c_void_p] * 4
    threads = effective_cpus()
    read_len = np.ascontiguousarray(ds.read_len, dtype=np.uint32)
    dev = (ctypes.c_int * world)(*devices)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        paf = os.path.join(d, "ovl.paf")
        ds.write_paf(paf)
        size = os.path.getsize(paf)
        best = None
        for _ in range(3):
            ms = [ctypes.c_double() for _ in range(2)]
            n_ovl, n_tr = ctypes.c_uint64(), ctypes.c_uint32()
            rc = L.rala_e2e_from_paf_ranks(paf.encode(), read_len.ctypes.data, ds.n_reads, threads, world, dev, 1 if transport == "local" else 0,
                                           *[ctypes.byref(x) for x in ms], ctypes.byref(n_ovl), ctypes.byref(n_tr))
            if rc != 0:
                raise RuntimeError("rala_e2e_from_paf_ranks: %d" % rc)
            tot = ms[0].value + ms[1].value
            if best is None or tot < best["ms_total"]:
                best = {"value": n_ovl.value / (tot * 1e-3), "unit": "overlaps/s", "threads": threads, "paf_bytes": size, "ranks": world,
                        "transport": transport, "ms_ingest": ms[0].value, "ms_device_first_call": ms[1].value, "ms_total": tot,
                        "transitive_pairs": n_tr.value, "ingest": "device tokeniser, every rank its own byte range of the file",
                        "source": "measured in this run (best of 3): name tables + text -> the ranks' devices -> tokenised there -> cuts between "
                                  "runs settled among the ranks -> sharded step"}
    return best


def stage_roofline(stage, n_ovl, sum_len, n_reads, ranks):
    b = (56.0 * n_ovl + 2.0 * sum_len + 40.0 * n_reads) / ranks
    ms = stage.get("dedupe_ms", 0.0) + stage.get("bucket_ms", 0.0) + stage.get("pile_ms", 0.0)
    ach = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"algorithmic_bytes": b, "ms": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBS}


def measure_traffic(workload):
    """HBM bytes of the pile kernel chain per step, measured HERE: two child runs of this command (one step, no baselines)
    under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` - separate passes, FETCH_SIZE doubled as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (KB units).  None when the profiler is not there or a pass
    fails: the line then replays profiles/pmc_latest.json and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    # the interpreter that runs this bench, as an ELF binary (a `python3` found on PATH may be a shim script
    # that execs the real one - an exec hop inside a process the profiler's preloaded library has initialised the GPU in -
    # or another interpreter altogether)
    py = os.path.realpath(sys.executable or "")
    try:
        with open(py, "rb") as f:
            is_elf = f.read(4) == b"\x7fELF"
    except OSError:
        is_elf = False
    if not os.path.exists(exe) or not is_elf:
        return None
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rala_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", RALA_BENCH_CHILD="1")
            # (the program itself behind `--`, nothing that re-executes: the profiler's library is in the process from the start)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", py,
                   os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--no-traffic",
                   "--workload", workload]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            total, seen = 0.0, False
            for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        name = row.get("Kernel_Name", "")
                        if row.get("Counter_Name") == counter and ("pile_runs_kernel" in name or "pile_build_annotate" in name):
                            total += float(row["Counter_Value"])
                            seen = True
            if not seen:
                return None
            kb[counter] = total
        except Exception as e:      # noqa: BLE001 - the line stands without it
            log("[bench] traffic pass %s failed: %s" % (counter, e))
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0


def gather_check_inputs(mine, valid_slice, row_fnv, row_sum, rank, world, n_reads):
    """one process per GPU: what result_check wants, on rank 0 - the replicated digests of rank 0 and of the last rank, the ranks'
    validity bytes behind each other (the slices are the file in order), the rows' checksums from their owners back in read order
    (entry j of rank k = read j * world + k).  Control plane (gloo); the other ranks get None."""
    import torch.distributed as dist
    box = [None] * world if rank == 0 else None
    dist.gather_object((mine, valid_slice, row_fnv, row_sum), box, dst=0)
    if rank != 0:
        return None
    fnv = np.zeros(n_reads, dtype=np.uint64)
    tot = np.zeros(n_reads, dtype=np.uint64)
    for k in range(world):
        fnv[k::world], tot[k::world] = box[k][2], box[k][3]
    return box[0][0], box[world - 1][0], np.concatenate([b[1] for b in box]), fnv, tot


def broadcast_flag(flag):
    """rank 0's verdict on every rank (one process per GPU)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    dist.broadcast(t, src=0)
    return bool(t.item())


RESULT_STAGES = ("valid", "piles2", "ov", "int", "nodes", "edges", "n_overlaps_kept", "n_internals_kept", "n_edges", "n_tr", "rows2", "rows2_sum")
RESULT_STAGES_SENS = ("valid", "piles3", "ov_sens", "rep", "nodes", "edges", "n_overlaps_kept_sens", "n_repeat_hills", "n_edges", "n_tr", "rows3", "rows3_sum")


def _dg(*arrays):
    """the digest of tests/golden/make_fullsize_digests.py"""
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def replicated_digests(ctx, n_tr, with_sens):
    """what every rank of a sharded run holds a copy of (and a single context holds once), behind the step's last call: read
    state, kept overlaps, graph with the transitive marks - as the digests tests/golden/fullsize_*.json keep of the reference
    objects' results"""
    out = {"n_tr": int(n_tr)}
    p = ctx.piles()
    lists = ("src", "a_begin", "a_end", "b_begin", "b_end", "length", "type")
    ov = ctx.overlap_list(0)
    if with_sens:
        out["piles3"] = _dg(*[p[k] for k in ("begin", "end", "median", "p10", "alive")])
        offs, pairs, flags = ctx.intervals(2)
        out["rep"] = _dg(offs.astype(np.uint64), pairs.astype(np.uint32), flags.astype(np.uint8))
        out["n_repeat_hills"] = int(len(pairs))
        out["n_overlaps_kept_sens"] = int(len(ov["src"]))
        out["ov_sens"] = _dg(*[np.asarray(ov[k]).astype(np.uint32) for k in lists])
    else:
        out["piles2"] = _dg(p["begin"], p["end"], p["alive"])
        it = ctx.overlap_list(1)
        out["n_overlaps_kept"], out["n_internals_kept"] = int(len(ov["src"])), int(len(it["src"]))
        out["ov"] = _dg(*[np.asarray(ov[k]).astype(np.uint32) for k in lists])
        out["int"] = _dg(*[np.asarray(it[k]).astype(np.uint32) for k in lists])
    g = ctx.graph()
    out["nodes"] = _dg(g["node_read"].astype(np.uint32))
    out["n_edges"] = int(len(g["src"]))
    out["edges"] = _dg(g["src"].astype(np.uint32), g["dst"].astype(np.uint32), g["len"].astype(np.uint32), g["marked"].astype(np.uint8))
    return out


def result_check(data_name, with_sens, first, last, valid, row_fnv, row_sum, ranks):
    """The timed path's RESULT against the committed digests of the reference objects' result on the same synthetic input
    (tests/golden/fullsize_<workload>[_sens].json, made by tests/golden/make_fullsize_digests.py): `first` / `last` =
    replicated_digests of rank 0 and of the last rank, `valid` = the validity bytes of all overlaps in file order (the ranks'
    slices behind each other), row_fnv / row_sum = the checksums of EVERY pile row from their owners, in read order.
    -> (the line's "result_check" member, ok).  Without a digest file for the workload (C5: no host here holds its reference
    objects) what is left to check is that the ranks agree."""
    path = os.path.join("tests", "golden", "fullsize_%s%s.json" % (data_name, "_sens" if with_sens else ""))
    got = dict(first)
    got["valid"] = _dg(np.packbits(valid))
    suffix = "3" if with_sens else "2"
    got["rows" + suffix] = _dg(row_fnv)
    got["rows" + suffix + "_sum"] = _dg(row_sum)
    if os.environ.get("RALA_BENCH_FAKE_WRONG_RESULT") == "1":       # (tests: a result that is not the reference's)
        got["edges"] = got["edges"][::-1]
    ranks_agree = all(first[k] == last[k] for k in first)
    out = {"digests": None, "ranks_checked": ranks, "ranks_agree": ranks_agree, "stages_equal": [], "stages_differ": [], "ok": None}
    want = None
    try:
        with open(os.path.join(ROOT, path)) as f:
            want = json.load(f)
    except OSError:
        pass
    if want is not None:
        out["digests"] = path
        out["backend"] = want.get("backend")
        for k in (RESULT_STAGES_SENS if with_sens else RESULT_STAGES):
            if k not in want:
                continue                    # (a digest file older than the stage)
            (out["stages_equal"] if got.get(k) == want[k] else out["stages_differ"]).append(k)
        out["ok"] = bool(ranks_agree and not out["stages_differ"] and len(out["stages_equal"]) >= 8)
    elif not ranks_agree:
        out["ok"] = False
    return out, out["ok"] is not False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("RALA_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end-from-PAF figure (it writes the data set as text first)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic here (two one-step child runs under rocprofv3 --pmc); replay profiles/pmc_latest.json")
    ap.add_argument("--no-result-check", action="store_true",
                    help="skip the comparison of the step's result with the committed digests (measurement scripts)")
    ap.add_argument("--transport", default=os.environ.get("RALA_COMM", "rccl"), choices=("rccl", "local"),
                    help="ranks as threads only: RCCL (default) or the in-process transport (peer copies)")
    ap.add_argument("--devices", default=os.environ.get("RALA_GPU_DEVICES", ""),
                    help="ranks as threads only: device ordinal per rank, comma separated (default 0,1,...)")
    args = ap.parse_args()

    import torch

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ                      # one process per GPU, started by torch.distributed.run
    force_sharded = os.environ.get("RALA_FORCE_SHARDED") == "1"
    use_dist = launched and (world_env > 1 or force_sharded)
    use_threads = not use_dist and (args.gpus > 1 or force_sharded)
    if launched and args.gpus != world_env:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d processes (WORLD_SIZE)" % (args.gpus, world_env))
    world = world_env if use_dist else args.gpus
    devices = [int(x) for x in args.devices.split(",") if x.strip() != ""] or list(range(world))
    rccl_failed = None
    if use_threads:
        # (no HIP call before the child below is started: where torch has no amdsmi this count goes through
        # hipGetDeviceCount, which is why the RCCL attempt is a fresh process and not a fork of this one)
        have = torch.cuda.device_count()
        if len(devices) != world:
            raise SystemExit("bench.py: --devices names %d devices for %d ranks" % (len(devices), world))
        if args.transport == "rccl" and (len(set(devices)) != world or max(devices) >= have):
            raise SystemExit("bench.py: --gpus %d needs %d distinct HIP devices, this node shows %d "
                             "(--transport local lets ranks share a device: a test of the decomposition, not a scaling run)"
                             % (world, world, have))
        if max(devices) >= have:
            raise SystemExit("bench.py: device %d does not exist (%d visible)" % (max(devices), have))
        # RCCL with more than one rank has never run on the boxes this was built on (one GPU each), and a collective that
        # hangs cannot be interrupted from inside the process: a bare `bench.py --gpus N` runs the RCCL ranks in a CHILD
        # process with a time limit.  The child's line is this run's line; a child that fails or hangs makes this run
        # FAIL (exit code 3, no line) - a figure that RCCL did not carry is never printed under this command
        # (`--transport local` has to be asked for by name).
        try_child = args.transport == "rccl" and "RALA_BENCH_CHILD" not in os.environ and \
            (world > 1 or os.environ.get("RALA_BENCH_TEST_CHILD") == "1")
        if try_child:
            import subprocess
            limit = int(os.environ.get("RALA_BENCH_CHILD_TIMEOUT", "900"))
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, RALA_BENCH_CHILD="1"),
                                   stdout=subprocess.PIPE, timeout=limit)
                lines = [x for x in r.stdout.decode(errors="replace").splitlines() if x.startswith("{")]
                if r.returncode == 0 and lines:
                    print(lines[-1], flush=True)
                    return
                rccl_failed = "exit code %d" % r.returncode
            except subprocess.TimeoutExpired:
                rccl_failed = "no result within %d s" % limit
            log("[bench] the RCCL run of %d ranks failed (%s).  No figure is printed: `--transport local` runs the same ranks "
                "over peer copies inside one process - a test of the decomposition, not the transport that was asked for."
                % (world, rccl_failed))
            raise SystemExit(3)
        if os.environ.get("RALA_BENCH_FAKE_RCCL_FAILURE") == "1" and "RALA_BENCH_CHILD" in os.environ and args.transport == "rccl":
            raise SystemExit(3)                     # (tests: the child of the paragraph above fails)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    if use_dist:
        import torch.distributed as dist
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no device %d" % (rank, local_rank))
        torch.cuda.set_device(local_rank)
        # control plane only (the RCCL id, the barrier, the max-over-ranks clock): gloo over loopback.
        # The data path is RCCL over xGMI, called from C++ on the contexts' streams.
        dist.init_process_group("gloo")

    from rala_amd import hip
    from rala_amd.synth import Dataset

    t0 = time.perf_counter()
    with_sens = args.workload.endswith("s")
    data_name = args.workload[:-1] if with_sens else args.workload
    ds = Dataset.config(data_name)
    n_ovl = len(ds.overlaps)
    sum_len = int(ds.read_len.astype(np.int64).sum())
    log("[bench] generated %s: %d reads, %d overlaps in %.1f s" % (args.workload, ds.n_reads, n_ovl,
                                                                    time.perf_counter() - t0))
    mode = "one GPU"
    transport, rccl_ranks = None, 0             # which transport carried the ranks (None: one context, nothing to carry)
    sens_info = None

    def sensitive_set(piles):
        """the sensitive overlaps of the two-run workflow, from the piles an (untimed) first pass leaves"""
        t1 = time.perf_counter()
        sens = ds.sensitive(piles["alive"], piles["begin"], piles["end"])
        targets = np.unique(sens.b_id)
        info = {"n_sensitive": len(sens), "targets": int(len(targets)),
                "sum_len_targets": int(ds.read_len[targets].astype(np.int64).sum())}
        log("[bench] %d sensitive overlaps on %d targets in %.1f s" % (len(sens), len(targets), time.perf_counter() - t1))
        return sens, info

    if use_dist:
        from rala_amd import multi
        try:
            runner = multi.ShardedRunner(ds, rank, world, local_rank)
        except Exception as e:      # noqa: BLE001 - no line without the transport that was asked for
            log("[bench] rank %d: the RCCL group of %d ranks could not be set up: %s" % (rank, world, e))
            raise SystemExit(3)
        mode = "one process per GPU (torch.distributed.run), RCCL"
        transport, rccl_ranks = "rccl", world
        if with_sens:
            runner.step()
            sens, sens_info = sensitive_set(runner.mg.context().piles())
            cut = [len(sens) * k // world for k in range(world + 1)]
            share = hip.DeviceOverlaps.from_host(sens.take(slice(cut[rank], cut[rank + 1])), local_rank)
            runner.mg.context().set_option("sensitive_in_device_memory", 1)
            def _step():
                n = runner.mg.run(share)
                runner._tm = multi.rank_timings(runner.mg)
                return n
            runner.step = _step
    elif use_threads:
        from rala_amd import multi
        try:
            runner = multi.ThreadedRunner(ds, world, devices, args.transport)
        except Exception as e:      # noqa: BLE001
            log("[bench] the %s group of %d ranks could not be set up: %s" % (args.transport, world, e))
            sys.stderr.flush()
            os._exit(3)             # (ranks may be stuck inside the library: no orderly exit)
        mode = "ranks as host threads of one process, %s" % ("RCCL" if args.transport == "rccl" else "in-process transport (peer copies)")
        transport, rccl_ranks = args.transport, (world if args.transport == "rccl" else 0)
        if with_sens:
            runner.step()
            sens, sens_info = sensitive_set(runner.ranks[0].context().piles())
            cut = [len(sens) * k // world for k in range(world + 1)]
            shares = [hip.DeviceOverlaps.from_host(sens.take(slice(cut[k], cut[k + 1])), devices[k]) for k in range(world)]
            for r in runner.ranks:
                r.context().set_option("sensitive_in_device_memory", 1)
            plain = runner.step
            runner.step = lambda: plain(shares)
    else:
        ctx = hip.Context(local_rank)
        for kv in filter(None, os.environ.get("RALA_BENCH_OPTIONS", "").split(",")):    # diagnostics: key=value,...
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        ctx.set_reads(ds.read_len)
        ctx.set_overlaps(ds.overlaps)          # inputs resident in HBM from here on
        sens_dev = None
        if with_sens:
            ctx.initialize()
            ctx.construct()
            sens, sens_info = sensitive_set(ctx.piles())
            sens_dev = hip.DeviceOverlaps.from_host(sens, local_rank)      # resident in HBM like the primary set
            ctx.set_option("sensitive_in_device_memory", 1)
            del sens

        class _Single:
            def step(self):
                ctx.initialize()
                ctx.construct(sens_dev)
                return ctx.remove_transitive_edges()

            def timings(self):
                return ctx.timings()
        runner = _Single()

    def barrier():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        elif use_threads:
            for d in sorted(set(devices)):
                torch.cuda.synchronize(d)
        else:
            torch.cuda.synchronize()

    n_tr = 0
    for _ in range(args.warmup):
        n_tr = runner.step()
    stage = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_tr = runner.step()
        for k, v in runner.timings().items():
            stage[k] = stage.get(k, 0.0) + float(v)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        from rala_amd import multi as _m
        dt = _m.max_over_ranks(dt)

    # ---- the result of the timed path against the committed digests (outside the timed region) ----
    check, check_ok = None, True
    if not args.no_result_check:
        if use_dist:
            mine = replicated_digests(runner.mg.context(), n_tr, with_sens) if rank in (0, world - 1) else None
            f, sm, _ = runner.mg.pile_row_digests()
            got = gather_check_inputs(mine, runner.mg.context().valid(), f, sm, rank, world, ds.n_reads)
            if rank == 0:
                check, check_ok = result_check(data_name, with_sens, *got, [0, world - 1])
            check_ok = bool(broadcast_flag(check_ok)) if world > 1 else check_ok
        elif use_threads:
            first = replicated_digests(runner.ranks[0].context(), n_tr, with_sens)
            last = replicated_digests(runner.ranks[world - 1].context(), n_tr, with_sens) if world > 1 else first
            fnv = np.zeros(ds.n_reads, dtype=np.uint64); tot = np.zeros(ds.n_reads, dtype=np.uint64)
            for k, r in enumerate(runner.ranks):
                fnv[k::world], tot[k::world], _ = r.pile_row_digests()
            check, check_ok = result_check(data_name, with_sens, first, last, np.concatenate([r.context().valid() for r in runner.ranks]),
                                           fnv, tot, [0, world - 1])
        else:
            first = replicated_digests(ctx, n_tr, with_sens)
            fnv, tot, _ = ctx.pile_row_digests()
            check, check_ok = result_check(data_name, with_sens, first, first, ctx.valid(), fnv, tot, [0])
        if not check_ok:
            if rank == 0:
                log("[bench] RESULT CHECK FAILED: %s" % json.dumps(check))
                log("[bench] the timed path did not produce the reference's result: no figure is printed")
            sys.stderr.flush()
            if use_threads:
                os._exit(4)
            raise SystemExit(4)

    if rank == 0:
        steps = max(1, args.steps)
        ms = 1000.0 * dt / steps
        for k in stage:
            stage[k] /= steps
        sharded = use_dist or use_threads
        # dominant kernel: the pile kernel chain (pile_runs_kernel).  Algorithmic bytes of one step's launches:
        # 16 B per overlap of bucketed bounds read + 2 B per base of pile written + 40 B per
        # read of annotations (DESIGN.md "Roofline"); time = HIP events
        # around those launches on the context's stream.
        pile_bytes = 16.0 * n_ovl + 2.0 * sum_len + 40.0 * ds.n_reads
        if sharded:
            pile_bytes /= world     # a rank's launches cover the reads it owns (1 / world of every term)
        pile_ms = stage.get("pile_ms", 0.0)
        achieved = pile_bytes / (pile_ms * 1e-3) / 1e9 if pile_ms > 0 else 0.0
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
                pm = json.load(f)
            if pm.get("workload") == args.workload:
                traffic = pm.get("hbm_bytes_per_step")      # replayed from the last PMC passes, not measured in this run
                if traffic is not None and sharded:
                    traffic /= world       # measured on one GPU over all reads; a rank's launches cover 1 / world
        except Exception:
            pass
        whole = stage_roofline(stage, n_ovl, sum_len, ds.n_reads, world if sharded else 1)
        if sens_info is not None:
            # the sensitive pass adds 32 B per sensitive overlap (ids, coordinates, its target bounds written
            # and read) and reads and writes the rows of its targets once more
            sens_info["algorithmic_bytes"] = 32.0 * sens_info["n_sensitive"] + 2.0 * sens_info["sum_len_targets"] * 2.0
            sens_info["ms"] = stage.get("repeats_ms", 0.0)
            if sens_info["ms"] > 0:
                sens_info["achieved_GBs"] = sens_info["algorithmic_bytes"] / (sens_info["ms"] * 1e-3) / 1e9 / (world if sharded else 1)
                sens_info["frac"] = sens_info["achieved_GBs"] / HBM_PEAK_GBS
            sens_info["resident"] = "device memory (option sensitive_in_device_memory), uploaded before the timed region"
        out = {
            "metric": "overlaps/sec (pile build + transitive reduction)",
            "value": n_ovl * steps / dt,
            "unit": "overlaps/s",
            "n_gpus": world,
            "transport": transport,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u16/u32 (+f64 compares)",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload], "n_reads": ds.n_reads, "n_overlaps": n_ovl,
                       "sum_read_len": sum_len, "transitive_pairs": int(n_tr), "ranks": mode},
            "roofline": {"bound": "hbm", "kernel": "pile_runs_kernel<512|1024|2048> + pile_build_annotate (overflow chain)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # the whole pile stage (dedupe + bucketing + pile kernels)
                         # against B_pile = 56 N_ovl + 2 sum(len) + 40 N_reads - next to the kernel's own figure
                         "stage_frac": whole["frac"], "stage_achieved": whole["achieved"], "stage_ms": whole["ms"],
                         "stage_algorithmic_bytes": whole["algorithmic_bytes"],
                         "traffic": traffic, "traffic_source": "REPLAYED from profiles/pmc_latest.json (rocprofv3 --pmc passes of an earlier run of this command; not re-measured here)" if traffic is not None else None,
                         "frac_of_achievable": achieved / ACHIEVABLE_GBS, "achievable": ACHIEVABLE_GBS,
                         "algorithmic_bytes": pile_bytes, "kernel_ms": pile_ms,
                         "stage": whole},
            "stage_ms": stage,
        }
        if sens_info is not None:
            out["sensitive_pass"] = sens_info
        if check is not None:
            out["result_check"] = check
        if not sharded:
            ctx.close()                 # the end-to-end run creates a context of its own: give the memory back first
        # (the full run only - what the driver launches; the quick runs of tests and measurement scripts pass --no-cpu-baseline -
        # and not inside a profiler)
        under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
        if (not args.no_traffic and not args.no_cpu_baseline and not sharded and world == 1 and not under_profiler and
                not os.environ.get("RALA_BENCH_CHILD")):
            measured = measure_traffic(args.workload)
            if measured is not None:
                out["roofline"]["traffic"] = measured
                out["roofline"]["traffic_source"] = ("measured in this run: two one-step child runs of this command under rocprofv3 --kernel-trace "
                                                     "--pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE doubled, gfx950), pile kernel chain")
        if not args.no_e2e and use_threads:
            runner.close()              # (the end-to-end run makes rank objects of its own)
            try:
                out["end_to_end_from_paf"] = end_to_end_from_paf_ranks(ds, world, devices, args.transport)
                assert out["end_to_end_from_paf"]["transitive_pairs"] == int(n_tr)
            except Exception as e:      # noqa: BLE001 - the headline figure stands without it
                log("[bench] end-to-end figure failed: %s" % e)
                out["end_to_end_from_paf"] = {"error": str(e)}
        if not args.no_e2e and world == 1 and not sharded and not with_sens:
            try:
                out["from_pinned_host"] = from_pinned_host(ds, local_rank)
                assert out["from_pinned_host"]["transitive_pairs"] == int(n_tr)
            except Exception as e:      # noqa: BLE001 - the headline figure stands without it
                log("[bench] host-fed figure failed: %s" % e)
                out["from_pinned_host"] = {"error": str(e)}
        if not args.no_e2e and world == 1 and not sharded:
            try:
                out["end_to_end_from_paf"] = end_to_end_from_paf(ds, args.workload)
            except Exception as e:      # noqa: BLE001 - the headline figure stands without it
                log("[bench] end-to-end figure failed: %s" % e)
                out["end_to_end_from_paf"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:
            # the N-thread figure on the configuration `value` is quoted on (C3 for the C5 workloads: what the box's host
            # memory and a few minutes hold), the one-thread figure on the C2 sample
            head = {"c1": "c1", "c2": "c2"}.get(data_name, "c3")
            out["cpu_baseline"] = cpu_baseline(head, "c1" if head == "c1" else "c2")
        print(json.dumps(out), flush=True)

    if use_dist or use_threads:
        runner.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
