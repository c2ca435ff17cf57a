"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/rala_hip.h declares.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rala_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rala_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from rala_amd import hip

    L = hip.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "librala_hip.so does not export %s" % n
    assert sorted(hip.SYMBOLS) == names, "rala_amd/hip.py SYMBOLS out of date with include/rala_hip.h"


def test_code_object_is_gfx950():
    from rala_amd import hip

    hip.lib()
    blob = open(hip.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"pile_build_annotate" in blob


def test_no_device_fails_loudly():
    """Without a usable HIP device the product path raises; it never falls back to the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rala_amd import hip

    with pytest.raises(hip.RalaHipError):
        hip.Context(0)


def test_product_does_not_touch_the_oracle():
    """nothing under rala_amd/ may import, link or execute oracle/"""
    pkg = os.path.join(ROOT, "rala_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if f == "build.py":
                    continue        # builds the checker for the tests; does not load it
                assert "oracle" not in text.lower(), "%s mentions the oracle" % os.path.join(base, f)


def test_reference_cli_compiles_unchanged():
    """SURVEY 8(b) "must build unchanged": the reference's src/main.cpp, compiled from where it
    lies, against this package's graph.hpp / sequence.hpp / thread_pool/thread_pool.hpp and
    linked with librala.so.  Only where /root/reference exists (this container)."""
    import subprocess

    from rala_amd import build

    if not os.path.exists(os.path.join(build.REFERENCE, "src", "main.cpp")):
        pytest.skip("no /root/reference here")
    build.build_host()
    exe = os.path.join(build.PKG, "host", "_refcli", "rala_ref")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "--version"], stdout=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.strip() == b"v1.0.0"
    ours = subprocess.run([os.path.join(build.PKG, "host", "rala"), "--version"], stdout=subprocess.PIPE)
    assert ours.returncode == 0 and ours.stdout.startswith(b"v1.0.0")
    # no file of the reference's text lives in the repository
    assert not os.path.exists(os.path.join(build.PKG, "host", "main.cpp"))


def test_thread_pool_header():
    """the product's thread_pool/thread_pool.hpp (interface of the reference's un-vendored
    submodule, graph.cpp:235,369): tasks run, futures deliver, the pool joins"""
    import subprocess
    import tempfile

    from rala_amd import build

    src = r'''
#include <stdio.h>
#include <numeric>
#include "thread_pool/thread_pool.hpp"
int main() {
    auto pool = thread_pool::createThreadPool(4);
    std::vector<std::future<long>> f;
    for (long i = 0; i < 1000; ++i) f.emplace_back(pool->submit_task([](long a, long b) { return a * b; }, i, 2L));
    long s = 0;
    for (auto& x : f) { x.wait(); s += x.get(); }
    auto v = pool->submit_task([]() {});
    v.wait();
    printf("%ld %u\n", s, pool->num_threads());
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-pthread", "-I" + os.path.join(build.PKG, "host"), "-o", exe,
                               os.path.join(d, "t.cpp")])
        out = subprocess.check_output([exe]).split()
    assert out == [b"999000", b"4"]


def test_create_sequence_keeps_the_reference_fatal_checks():
    """rala::createSequence leaves the process with the reference's message on an empty name / data (reference
    src/sequence.cpp:15-22; SURVEY 8(b): error texts kept).  CPU only: the factory touches no device."""
    import subprocess
    import sys

    lib = os.path.join(ROOT, "rala_amd", "host", "librala_api.so")
    from rala_amd import build
    build.build_host()
    code = ("import ctypes, sys; L = ctypes.CDLL(%r); L.hp_sequence_length.restype = ctypes.c_uint64; "
            "L.hp_sequence_length.argtypes = [ctypes.c_char_p, ctypes.c_char_p]; "
            "print(L.hp_sequence_length(sys.argv[1].encode(), sys.argv[2].encode()))" % lib)
    ok = subprocess.run([sys.executable, "-c", code, "read1", "ACGT"], capture_output=True, text=True)
    assert ok.returncode == 0 and ok.stdout.strip() == "4", ok.stderr
    for name, data, what in (("", "ACGT", "name"), ("read1", "", "data")):
        res = subprocess.run([sys.executable, "-c", code, name, data], capture_output=True, text=True)
        assert res.returncode == 1
        assert "[rala::createSequence] error: empty %s!" % what in res.stderr
