"""rala_amd.cpus.effective_cpus: affinity mask cut by the cgroup CPU quota."""
import builtins
import io
import os

from rala_amd import cpus


def test_effective_cpus_is_positive_and_bounded():
    n = cpus.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_quota_is_honoured(monkeypatch):
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return io.StringIO("250000 100000\n")
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)))
    assert cpus.effective_cpus() == 3           # 2.5 CPUs' worth of time, rounded up

    def unlimited(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return io.StringIO("max 100000\n")
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", unlimited)
    assert cpus.effective_cpus() == 64
