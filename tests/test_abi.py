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
