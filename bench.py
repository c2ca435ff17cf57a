#!/usr/bin/env python3
"""Throughput of Rala's hot path on MI355X: overlaps/sec for pile build + filtering +
graph construction + transitive reduction (BASELINE.json metric), synthetic input of the
named shape resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c1|c5]

One step = one pass of the whole hot path over the data set: rala_hip_initialize
(duplicate removal, bound bucketing, pile build + annotation), rala_hip_construct
(second overlap pass, containment fixed point, preprocess tail, graph build) and
rala_hip_remove_transitive_edges.  Prints ONE JSON line on rank 0.

Several GPUs (reads hash-partitioned, ONE all-to-all of bound tuples; rala_hip_mg_*):
  * bare `python bench.py --gpus N`: the N ranks are host threads of this process, one per
    GPU, over RCCL - what `rala --gpus N` does (rala_amd/host/graph.cpp);
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: one
    process per GPU; torch carries only the 128-byte RCCL id, the barrier and the clock (gloo).
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

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
WRITE_ONLY_GBS = 5100.0 # what a kernel that only writes the same rows reaches (tools/fill_bench.hip)

WORKLOADS = {
    "c1": "1k reads / 50k overlaps (BASELINE configs[0])",
    "c2": "100k reads / 5M overlaps (BASELINE configs[1])",
    "c3": "1M reads / 50M overlaps (BASELINE configs[2])",
    "c5": "4M reads / 300M overlaps (BASELINE configs[4], primary overlaps only)",
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(sample_name="c2"):
    """The CPU path on a bounded sample: one thread AND as many threads as the process may really
    use, on the SAME sample, one task per pile like the reference's thread pool.  kind "reference":
    the reference's own rala::Pile / rala::Overlap objects (oracle/_ref, compiled from the
    reference's pile.cpp / overlap.cpp) under the restated Graph orchestration; kind "port": the
    flat restatement (oracle/_build) when that library is not there.  Both are test infrastructure
    used here only as the measured baseline."""
    from oracle import oracle as ora
    from rala_amd.synth import Dataset

    from rala_amd.cpus import effective_cpus

    cores = effective_cpus()        # affinity mask cut by the container's CPU quota (16 on the gpurun boxes)
    ds = Dataset.config(sample_name)
    n = len(ds.overlaps)

    def once(ref, threads):
        t0 = time.perf_counter()
        o = ora.Oracle(ds.read_len, ds.overlaps, n_threads=threads, ref=ref)
        rc = o.construct()
        n_tr = o.remove_transitive_edges() if rc == 0 else 0
        return time.perf_counter() - t0, n_tr

    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liboracle_ref.so"))
    out = {"unit": "overlaps/s", "cores": cores}
    kind = "port"
    dt_port, n_tr = once(False, cores)
    dt = dt_port
    if have_ref:
        try:
            dt_ref, n_tr_ref = once(True, cores)
            assert n_tr_ref == n_tr
            kind, dt = "reference", dt_ref
            out["port_value"] = n / dt_port
        except Exception as e:          # the library is there but unusable on this box: say so, fall back
            log("[bench] reference-object baseline failed (%s); using the port" % e)
    dt1, n_tr1 = once(kind == "reference", 1)
    assert n_tr1 == n_tr
    out.update(value=n / dt, kind=kind, value_1_thread=n / dt1,
               sample="%s synthetic, %d reads / %d overlaps, whole hot path once with %s: %.2f s on %d threads, %.2f s on "
                      "one thread (same sample), %d transitive pairs" % (
                          sample_name, ds.n_reads, n, "the reference's Pile / Overlap objects" if kind == "reference"
                          else "the flat restatement", dt, cores, dt1, n_tr))
    return out


def end_to_end_from_paf(ds, workload):
    """The second figure of SURVEY 8(d), measured in THIS run: the data set written as PAF text to a
    scratch file, then multi-threaded ingest + upload + one pass of the device path
    (rala_e2e_from_paf in librala.so; best of three)."""
    from rala_amd import build
    from rala_amd.cpus import effective_cpus

    build.build_host()
    L = ctypes.CDLL(os.path.join(build.PKG, "host", "librala.so"))
    L.rala_e2e_from_paf.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32] + [ctypes.c_void_p] * 5
    threads = effective_cpus()
    read_len = np.ascontiguousarray(ds.read_len, dtype=np.uint32)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        paf = os.path.join(d, "ovl.paf")
        t0 = time.perf_counter()
        ds.write_paf(paf)
        size = os.path.getsize(paf)
        log("[bench] end to end: wrote %.2f GB of PAF in %.1f s" % (size / 1e9, time.perf_counter() - t0))
        best = None
        for _ in range(3):
            ms = [ctypes.c_double() for _ in range(3)]
            n_ovl, n_tr = ctypes.c_uint64(), ctypes.c_uint32()
            rc = L.rala_e2e_from_paf(paf.encode(), read_len.ctypes.data, ds.n_reads, threads, *[ctypes.byref(x) for x in ms],
                                     ctypes.byref(n_ovl), ctypes.byref(n_tr))
            if rc != 0:
                raise RuntimeError("rala_e2e_from_paf: %d" % rc)
            tot = sum(x.value for x in ms)
            if best is None or tot < best["ms_total"]:
                best = {"value": n_ovl.value / (tot * 1e-3), "unit": "overlaps/s", "threads": threads, "paf_bytes": size,
                        "ms_parse": ms[0].value, "ms_upload": ms[1].value, "ms_device_first_call": ms[2].value,
                        "ms_total": tot, "transitive_pairs": n_tr.value,
                        "source": "measured in this run (best of 3): PAF text -> threaded ingest -> upload -> device path"}
    return best


def stage_roofline(stage, n_ovl, sum_len, n_reads, ranks):
    b = (56.0 * n_ovl + 2.0 * sum_len + 40.0 * n_reads) / ranks
    ms = stage.get("dedupe_ms", 0.0) + stage.get("bucket_ms", 0.0) + stage.get("pile_ms", 0.0)
    ach = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"algorithmic_bytes": b, "ms": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("RALA_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end-from-PAF figure (it writes the data set as text first)")
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
        have = torch.cuda.device_count()            # (does not initialise the GPU: a child process may still be started)
        if len(devices) != world:
            raise SystemExit("bench.py: --devices names %d devices for %d ranks" % (len(devices), world))
        if args.transport == "rccl" and (len(set(devices)) != world or max(devices) >= have):
            raise SystemExit("bench.py: --gpus %d needs %d distinct HIP devices, this node shows %d "
                             "(--transport local lets ranks share a device: a test of the decomposition, not a scaling run)"
                             % (world, world, have))
        if max(devices) >= have:
            raise SystemExit("bench.py: device %d does not exist (%d visible)" % (max(devices), have))
        # RCCL with more than one rank has never run on the boxes this was built on (one GPU each).  A bare
        # `bench.py --gpus N` therefore tries it in a CHILD process first and, should that fail or hang, runs the same
        # ranks over the in-process transport (peer copies between the devices) - and says so in the line.
        try_child = args.transport == "rccl" and "RALA_BENCH_CHILD" not in os.environ and \
            (world > 1 or os.environ.get("RALA_BENCH_TEST_CHILD") == "1") and os.environ.get("RALA_BENCH_NO_FALLBACK") != "1"
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
            log("[bench] the RCCL run failed (%s): the same ranks over the in-process transport" % rccl_failed)
            args.transport = "local"
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
    ds = Dataset.config(args.workload)
    n_ovl = len(ds.overlaps)
    sum_len = int(ds.read_len.astype(np.int64).sum())
    log("[bench] generated %s: %d reads, %d overlaps in %.1f s" % (args.workload, ds.n_reads, n_ovl,
                                                                    time.perf_counter() - t0))
    mode = "one GPU"
    if use_dist:
        from rala_amd import multi
        runner = multi.ShardedRunner(ds, rank, world, local_rank)
        mode = "one process per GPU (torch.distributed.run), RCCL"
    elif use_threads:
        from rala_amd import multi
        runner = multi.ThreadedRunner(ds, world, devices, args.transport)
        mode = "ranks as host threads of one process, %s" % ("RCCL" if args.transport == "rccl" else "in-process transport (peer copies)")
        if rccl_failed:
            mode += " - the RCCL run failed: " + rccl_failed
    else:
        ctx = hip.Context(local_rank)
        for kv in filter(None, os.environ.get("RALA_BENCH_OPTIONS", "").split(",")):    # diagnostics: key=value,...
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        ctx.set_reads(ds.read_len)
        ctx.set_overlaps(ds.overlaps)          # inputs resident in HBM from here on

        class _Single:
            def step(self):
                ctx.initialize()
                ctx.construct()
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
        out = {
            "metric": "overlaps/sec (pile build + transitive reduction)",
            "value": n_ovl * steps / dt,
            "unit": "overlaps/s",
            "n_gpus": world,
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
                         "frac_of_write_only_rate": achieved / WRITE_ONLY_GBS,
                         "algorithmic_bytes": pile_bytes, "kernel_ms": pile_ms,
                         "stage": whole},
            "stage_ms": stage,
        }
        if not sharded:
            ctx.close()                 # the end-to-end run creates a context of its own: give the memory back first
        if not args.no_e2e and world == 1 and not sharded:
            try:
                out["end_to_end_from_paf"] = end_to_end_from_paf(ds, args.workload)
            except Exception as e:      # noqa: BLE001 - the headline figure stands without it
                log("[bench] end-to-end figure failed: %s" % e)
                out["end_to_end_from_paf"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline("c2" if args.workload != "c1" else "c1")
        print(json.dumps(out), flush=True)

    if use_dist or use_threads:
        runner.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
