#!/usr/bin/env python3
"""Throughput of Rala's hot path on MI355X: overlaps/sec for pile build + filtering +
graph construction + transitive reduction (BASELINE.json metric), synthetic input of the
named shape resident in HBM before the timed region.

    python bench.py --gpus 1 --steps K --warmup W [--workload c3|c2|c1]

One step = one pass of the whole hot path over the data set: rala_hip_initialize
(duplicate removal, bound bucketing, pile build + annotation), rala_hip_construct
(second overlap pass, containment fixed point, preprocess tail, graph build) and
rala_hip_remove_transitive_edges.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
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
    """The CPU path on a bounded sample, as many threads as the process may really use, one
    task per pile like the reference's thread pool.  kind "reference": the reference's own rala::Pile / rala::Overlap objects
    (oracle/_ref, compiled from the reference's pile.cpp / overlap.cpp) under the restated
    Graph orchestration; kind "port": the flat restatement (oracle/_build) when that library
    is not there.  Both are test infrastructure used here only as the measured baseline."""
    from oracle import oracle as ora
    from rala_amd.synth import Dataset

    from rala_amd.cpus import effective_cpus

    cores = effective_cpus()        # affinity mask cut by the container's CPU quota (16 on the gpurun boxes)
    ds = Dataset.config(sample_name)

    def once(ref):
        t0 = time.perf_counter()
        o = ora.Oracle(ds.read_len, ds.overlaps, n_threads=cores, ref=ref)
        rc = o.construct()
        n_tr = o.remove_transitive_edges() if rc == 0 else 0
        return time.perf_counter() - t0, n_tr

    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liboracle_ref.so"))
    dt_port, n_tr = once(False)
    out = {"unit": "overlaps/s", "cores": cores}
    # one thread, on a smaller sample (a fifth of the reads at the same coverage) so that it stays seconds
    small = Dataset(ds.n_reads // 5, max(1, int(ds.read_len.astype(np.int64).sum() // 250)), 7) if ds.n_reads >= 5000 else ds
    t0 = time.perf_counter()
    o1 = ora.Oracle(small.read_len, small.overlaps, n_threads=1, ref=have_ref)
    if o1.construct() == 0:
        o1.remove_transitive_edges()
    out["value_1_thread"] = len(small.overlaps) / (time.perf_counter() - t0)
    out["sample_1_thread"] = "%d reads / %d overlaps" % (small.n_reads, len(small.overlaps))
    if have_ref:
        try:
            dt_ref, n_tr_ref = once(True)
            assert n_tr_ref == n_tr
            out.update(value=len(ds.overlaps) / dt_ref, kind="reference", port_value=len(ds.overlaps) / dt_port,
                       sample="%s synthetic, %d reads / %d overlaps, whole hot path once: %.2f s with the reference's "
                              "Pile / Overlap objects, %.2f s with the flat restatement, %d transitive pairs" % (
                                  sample_name, ds.n_reads, len(ds.overlaps), dt_ref, dt_port, n_tr))
            return out
        except Exception as e:          # the library is there but unusable on this box: say so, fall back
            log("[bench] reference-object baseline failed (%s); using the port" % e)
    out.update(value=len(ds.overlaps) / dt_port, kind="port",
               sample="%s synthetic, %d reads / %d overlaps, whole hot path once, %.2f s, %d transitive pairs" % (
                   sample_name, ds.n_reads, len(ds.overlaps), dt_port, n_tr))
    return out


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
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_sharded = os.environ.get("RALA_FORCE_SHARDED") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_sharded
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d - launch one process per GPU:\n"
                         "  python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 "
                         "--master-port 29500 bench.py --gpus %d ..." % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    if use_dist:
        import torch.distributed as dist
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
    if use_dist:
        from rala_amd import multi
        runner = multi.ShardedRunner(ds, rank, world, local_rank)
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
        # dominant kernel: the pile kernel chain (pile_runs_kernel).  Algorithmic bytes of one step's launches:
        # 16 B per overlap of bucketed bounds read + 2 B per base of pile written + 40 B per
        # read of annotations (DESIGN.md "Roofline"); time = HIP events
        # around those launches on the context's stream.
        pile_bytes = 16.0 * n_ovl + 2.0 * sum_len + 40.0 * ds.n_reads
        if use_dist:
            pile_bytes /= world     # a rank's launches cover the reads it owns (1 / world of every term)
        pile_ms = stage.get("pile_ms", 0.0)
        achieved = pile_bytes / (pile_ms * 1e-3) / 1e9 if pile_ms > 0 else 0.0
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
                pm = json.load(f)
            if pm.get("workload") == args.workload:
                traffic = pm.get("hbm_bytes_per_step")      # replayed from the last PMC passes, not measured in this run
                if traffic is not None and use_dist:
                    traffic /= world       # measured on one GPU over all reads; a rank's launches cover 1 / world
        except Exception:
            pass
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
                       "sum_read_len": sum_len, "transitive_pairs": int(n_tr)},
            "roofline": {"bound": "hbm", "kernel": "pile_runs_kernel<512|1024|2048> + pile_build_annotate (overflow chain)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": "profiles/pmc_latest.json (rocprofv3 --pmc passes of an earlier run of this command; not re-measured here)" if traffic is not None else None,
                         "frac_of_write_only_rate": achieved / WRITE_ONLY_GBS,
                         "algorithmic_bytes": pile_bytes, "kernel_ms": pile_ms,
                         # the whole pile stage (dedupe + bucketing + pile kernels)
                         # against B_pile = 56 N_ovl + 2 sum(len) + 40 N_reads
                         "stage": stage_roofline(stage, n_ovl, sum_len, ds.n_reads, world if use_dist else 1)},
            "stage_ms": stage,
        }
        # the second figure of SURVEY 8(d), from PAF text (ingest + upload + device): measured by
        # tools/e2e_bench.py (it writes a 3 GB file first), replayed here from its last committed run
        try:
            with open(os.path.join(ROOT, "profiles", "r02_%s_e2e_from_paf.json" % args.workload)) as f:
                e2e = json.load(f)
            out["end_to_end_from_paf"] = {"value": e2e["overlaps_per_s"], "unit": "overlaps/s", "threads": e2e["threads"],
                                          "ms_parse": e2e["ms_parse"], "ms_upload": e2e["ms_upload"],
                                          "source": "profiles/r02_%s_e2e_from_paf.json (tools/e2e_bench.py; not re-measured in this run)" % args.workload}
        except Exception:
            pass
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline("c2" if args.workload != "c1" else "c1")
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
