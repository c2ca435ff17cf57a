"""CPU: bench.py's result check (the comparison of a step's result with the committed digests of the reference objects' result,
tests/golden/fullsize_*.json) - the logic alone, on made-up digests; the GPU suite runs it on real results (test_gpu_bench.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _inputs(want, with_sens=False):
    import bench

    stages = bench.RESULT_STAGES_SENS if with_sens else bench.RESULT_STAGES
    first = {k: want[k] for k in stages if k in want and not k.startswith(("valid", "rows"))}
    return first


def test_result_check_against_a_committed_file(tmp_path, monkeypatch):
    import bench

    want = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_c2.json")))
    assert want["backend"] == "reference-objects"
    for k in ("rows0", "rows0_sum", "rows2", "rows2_sum"):
        assert k in want, "tests/golden/fullsize_c2.json lacks the digest of every pile row (%s)" % k
    first = _inputs(want)
    valid = np.zeros(16, dtype=np.uint8)
    rows = np.zeros(4, dtype=np.uint64)
    # the replicated stages equal, the made-up validity bytes and row checksums not: the check fails and says where
    out, ok = bench.result_check("c2", False, first, dict(first), valid, rows, rows, [0, 7])
    assert not ok and out["ok"] is False and out["ranks_agree"]
    assert set(out["stages_differ"]) == {"valid", "rows2", "rows2_sum"}
    assert set(out["stages_equal"]) == set(first)
    assert out["digests"] == "tests/golden/fullsize_c2.json" and out["ranks_checked"] == [0, 7]
    # two ranks that disagree fail whatever the file says
    last = dict(first, edges="0" * 64)
    out, ok = bench.result_check("c2", False, first, last, valid, rows, rows, [0, 1])
    assert not ok and not out["ranks_agree"]
    # a workload without a digest file: nothing to compare with, agreement of the ranks is all there is
    out, ok = bench.result_check("c5", False, first, dict(first), valid, rows, rows, [0, 7])
    assert ok and out["ok"] is None and out["digests"] is None and out["ranks_agree"]
    out, ok = bench.result_check("c5", False, first, last, valid, rows, rows, [0, 7])
    assert not ok and out["ok"] is False


def test_result_check_passes_on_the_committed_digests(monkeypatch):
    """every stage as the file has it: ok (the validity / row digests injected through the digest function)"""
    import bench

    want = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_c3.json")))
    first = _inputs(want)
    answers = iter([want["valid"], want["rows2"], want["rows2_sum"]])
    monkeypatch.setattr(bench, "_dg", lambda *a: next(answers))
    out, ok = bench.result_check("c3", False, first, dict(first), np.zeros(8, np.uint8), np.zeros(1, np.uint64), np.zeros(1, np.uint64), [0, 7])
    assert ok and out["ok"] is True and not out["stages_differ"] and len(out["stages_equal"]) == len(bench.RESULT_STAGES)
