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


def test_rccl_attempt_in_a_child_and_the_fallback():
    """A bare `--gpus N` tries RCCL in a child process first (more than one RCCL rank has never run where this was
    built); a child that fails sends the same ranks through the in-process transport, and the line says so.  Here with
    a world of one (RALA_FORCE_SHARDED), the child forced by RALA_BENCH_TEST_CHILD."""
    common = ("--workload", "c1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e")
    ok = bench(*common, env={"RALA_FORCE_SHARDED": "1", "RALA_BENCH_TEST_CHILD": "1"})
    assert ok.returncode == 0, ok.stderr[-2000:]
    a = json.loads(ok.stdout.strip().splitlines()[-1])
    assert "RCCL" in a["config"]["ranks"] and "failed" not in a["config"]["ranks"]
    bad = bench(*common, env={"RALA_FORCE_SHARDED": "1", "RALA_BENCH_TEST_CHILD": "1", "RALA_BENCH_FAKE_RCCL_FAILURE": "1"})
    assert bad.returncode == 0, bad.stderr[-2000:]
    b = json.loads(bad.stdout.strip().splitlines()[-1])
    assert "in-process transport" in b["config"]["ranks"] and "the RCCL run failed: exit code 3" in b["config"]["ranks"]
    assert b["config"]["transitive_pairs"] == a["config"]["transitive_pairs"]
