"""GPU: bench.py's launch modes on one device - the bare single-GPU line, bare `--gpus N` (ranks as
host threads of the bench process; here N ranks share device 0 through the in-process transport,
which is what a one-GPU box allows), and the refusal of an RCCL run with fewer devices than ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=e, timeout=600)
    return res


def test_bare_bench_runs_one_and_several_ranks():
    one = bench("--workload", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e")
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    assert a["n_gpus"] == 1 and a["roofline"]["frac"] > 0 and a["roofline"]["stage_frac"] > 0
    many = bench("--gpus", "4", "--transport", "local", "--devices", "0,0,0,0", "--workload", "c2", "--steps", "2",
                 "--warmup", "1", "--no-cpu-baseline", "--no-e2e")
    assert many.returncode == 0, many.stderr[-2000:]
    b = json.loads(many.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 4 and b["scaling"] == "strong"
    assert "threads" in b["config"]["ranks"]
    assert b["config"]["transitive_pairs"] == a["config"]["transitive_pairs"]
    assert b["config"]["n_overlaps"] == a["config"]["n_overlaps"]
    assert b["stage_ms"]["exchange_ms"] > 0 and b["roofline"]["kernel_ms"] > 0
    # the step's RESULT against the committed digests of the reference objects' result, in the line (round 6): validity bytes,
    # read state, kept overlaps, graph with its transitive marks and the checksum of EVERY pile row - rank 0 and the last rank
    for line, ranks in ((a, [0]), (b, [0, 3])):
        rc = line["result_check"]
        assert rc["ok"] is True and rc["digests"] == "tests/golden/fullsize_c2.json" and rc["backend"] == "reference-objects"
        assert rc["ranks_checked"] == ranks and rc["ranks_agree"] and not rc["stages_differ"]
        assert set(rc["stages_equal"]) >= {"valid", "piles2", "ov", "int", "nodes", "edges", "n_tr", "rows2", "rows2_sum"}
    assert "frac_of_achievable" in a["roofline"] and "frac_of_write_only_rate" not in a["roofline"]


def test_a_wrong_result_prints_no_figure():
    """a step whose result differs from the committed digests ends with exit code 4 and no line (RALA_BENCH_FAKE_WRONG_RESULT
    turns one digest over before the comparison)"""
    for extra in ((), ("--gpus", "2", "--transport", "local", "--devices", "0,0")):
        res = bench("--workload", "c2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", *extra,
                    env={"RALA_BENCH_FAKE_WRONG_RESULT": "1"})
        assert res.returncode == 4, (res.returncode, res.stderr[-2000:])
        assert not [x for x in res.stdout.splitlines() if x.startswith("{")]
        assert "RESULT CHECK FAILED" in res.stderr and "edges" in res.stderr


def test_bench_refuses_more_rccl_ranks_than_devices():
    import torch

    n = torch.cuda.device_count() + 1
    res = bench("--gpus", str(n), "--workload", "c1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e")
    assert res.returncode != 0
    assert "distinct HIP devices" in (res.stderr + res.stdout)


def test_bench_end_to_end_figure_is_measured():
    res = bench("--workload", "c1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    e2e = line["end_to_end_from_paf"]
    assert e2e["value"] > 0 and "measured in this run" in e2e["source"]
    assert e2e["transitive_pairs"] == line["config"]["transitive_pairs"]


def test_rccl_attempt_in_a_child_and_no_silent_fallback():
    """A bare `--gpus N` runs the RCCL ranks in a child process with a time limit (more than one RCCL rank has never run
    where this was built, and a hung collective cannot be interrupted from inside); the child's line is the run's line, and
    it names the transport.  A child that fails makes the run FAIL - round 3 printed a figure carried by peer copies
    instead.  Here with a world of one (RALA_FORCE_SHARDED), the child forced by RALA_BENCH_TEST_CHILD."""
    common = ("--workload", "c1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e")
    ok = bench(*common, env={"RALA_FORCE_SHARDED": "1", "RALA_BENCH_TEST_CHILD": "1"})
    assert ok.returncode == 0, ok.stderr[-2000:]
    a = json.loads(ok.stdout.strip().splitlines()[-1])
    assert a["transport"] == "rccl" and a["rccl_ranks"] == a["n_gpus"] == 1 and "RCCL" in a["config"]["ranks"]
    bad = bench(*common, env={"RALA_FORCE_SHARDED": "1", "RALA_BENCH_TEST_CHILD": "1", "RALA_BENCH_FAKE_RCCL_FAILURE": "1"})
    assert bad.returncode == 3, (bad.returncode, bad.stderr[-2000:])
    assert not [x for x in bad.stdout.splitlines() if x.startswith("{")]
    assert "--transport local" in bad.stderr
    # the in-process transport, asked for by name, says what it is
    loc = bench(*common, "--transport", "local", env={"RALA_FORCE_SHARDED": "1"})
    assert loc.returncode == 0, loc.stderr[-2000:]
    b = json.loads(loc.stdout.strip().splitlines()[-1])
    assert b["transport"] == "local" and b["rccl_ranks"] == 0
    assert b["config"]["transitive_pairs"] == a["config"]["transitive_pairs"]


def test_a_rank_without_a_device_fails_the_set_up_at_once():
    """every rank's device contexts are created before anybody joins the group (rala_hip_mg_create_contexts / _join): a rank
    that has no device ends the run with an error instead of leaving the others inside ncclCommInitRank"""
    import time

    t0 = time.time()
    res = bench("--gpus", "2", "--transport", "local", "--devices", "0,0", "--workload", "c1", "--steps", "1", "--warmup", "0",
                "--no-cpu-baseline", "--no-e2e", env={"RALA_HIP_DEBUG_FAIL_RANK": "1"})
    assert res.returncode == 3, (res.returncode, res.stderr[-2000:])
    assert "device contexts" in res.stderr and time.time() - t0 < 120


def test_sensitive_workload():
    """c2s: the sensitive second pass inside the timed step, its overlaps resident in HBM; the line carries the pass' own
    bytes and time; the same over two ranks that share the device"""
    one = bench("--workload", "c2s", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e")
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    sp = a["sensitive_pass"]
    assert sp["n_sensitive"] > 100_000 and sp["ms"] > 0 and sp["algorithmic_bytes"] > 0 and a["transport"] is None
    plain = bench("--workload", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e")
    p = json.loads(plain.stdout.strip().splitlines()[-1])
    assert a["config"]["transitive_pairs"] != p["config"]["transitive_pairs"]          # the pass drops overlaps
    two = bench("--gpus", "2", "--transport", "local", "--devices", "0,0", "--workload", "c2s", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline", "--no-e2e")
    assert two.returncode == 0, two.stderr[-2000:]
    b = json.loads(two.stdout.strip().splitlines()[-1])
    assert b["config"]["transitive_pairs"] == a["config"]["transitive_pairs"]
    assert b["sensitive_pass"]["n_sensitive"] == sp["n_sensitive"] and b["sensitive_pass"]["ms"] > 0
    for line in (a, b):
        rc = line["result_check"]
        assert rc["ok"] is True and rc["digests"] == "tests/golden/fullsize_c2_sens.json" and not rc["stages_differ"]
        assert set(rc["stages_equal"]) >= {"valid", "piles3", "ov_sens", "rep", "nodes", "edges", "n_tr", "rows3"}


def test_one_process_per_gpu_launch():
    """the driver's launch for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), here with N = 1 and the
    sharded runner forced: every rank's device contexts first, then the RCCL join (rala_amd/multi.py: ShardedRunner), the
    step through RCCL - with and without the sensitive pass (its share resident on the rank's device)"""
    e = dict(os.environ, RALA_FORCE_SHARDED="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    lines = {}
    for wl in ("c2", "c2s"):
        res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                              "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", wl,
                              "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        lines[wl] = json.loads([x for x in res.stdout.splitlines() if x.startswith("{")][-1])
    a, b = lines["c2"], lines["c2s"]
    assert a["transport"] == "rccl" and a["rccl_ranks"] == 1 and a["n_gpus"] == 1 and a["stage_ms"]["exchange_ms"] > 0
    assert b["sensitive_pass"]["n_sensitive"] > 100_000 and b["sensitive_pass"]["ms"] > 0
    assert b["config"]["transitive_pairs"] != a["config"]["transitive_pairs"]
    assert a["result_check"]["ok"] is True and b["result_check"]["ok"] is True
    assert a["result_check"]["digests"] == "tests/golden/fullsize_c2.json" and b["result_check"]["digests"] == "tests/golden/fullsize_c2_sens.json"


def test_traffic_is_measured_in_the_run():
    """roofline.traffic of the full bench run: two one-step child runs of bench.py under rocprofv3 --pmc (bench.measure_traffic),
    the pile kernels' FETCH_SIZE (doubled, gfx950) + WRITE_SIZE.  C2: 100 k reads of 1 Gbase, 5.09 M overlaps - 2.09 GB
    algorithmic (16 B per overlap + 2 B per base + 40 B per read)."""
    import shutil

    sys.path.insert(0, ROOT)
    import bench

    assert shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"), "no rocprofv3 on this box"
    t = bench.measure_traffic("c2")
    assert t is not None, "the PMC passes failed"
    assert 1.9e9 < t < 2.8e9, t
