"""CPU tests: the C-ABI library loads, exports every symbol include/corb_accel.h declares, and fails
loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "corb_accel.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(corb_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(corb):
    L = corb.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libcorb_accel.so does not export %s" % n
    assert set(corb.EXPORTS) == set(names)


def test_version_and_struct_sizes(corb):
    assert corb.load().corb_version() >= 100
    assert corb.KP_DTYPE.itemsize == 28 and corb.EDGE_DTYPE.itemsize == 24
    assert ctypes.sizeof(corb.OrbConfig) == 36


def test_no_cpu_fallback_without_device(corb):
    """Without a GPU every compute entry point must fail loudly."""
    if corb.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(corb.CorbError):
        corb.ORBextractor()
    with pytest.raises(corb.CorbError):
        corb.StereoFrontend()
    a = np.zeros((4, 32), np.uint8)
    with pytest.raises(corb.CorbError):
        corb.ORBmatcher.DescriptorDistance(a, a)


def test_product_does_not_reference_oracle():
    """The product tree must not import / link / name the oracle."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "corb-slam_amd")):
        if "build" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"liborc|pyorc|from oracle|import oracle|oracle/orc", txt):
                    if f == "synth.py":
                        continue
                    bad.append(os.path.join(dp, f))
    # comments may cite the oracle as the parity partner, but nothing may load it
    for p in bad:
        txt = open(p, errors="replace").read()
        assert not re.search(r"(dlopen|CDLL|import|#include)[^\n]*(liborc|pyorc|oracle)", txt), p
