"""GPU test: the C++ host program (reference-shaped classes over the C-ABI) reproduces the oracle."""
import os
import subprocess
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_host_program(tmp_path, pyorc, synth):
    exe = tmp_path / "host_smoke"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "corb-slam_amd", "host"),
                           os.path.join(ROOT, "corb-slam_amd", "host", "host_smoke.cpp"), "-o", str(exe), "-L", os.path.join(ROOT, "corb-slam_amd"),
                           "-lcorb_accel", "-Wl,-rpath," + os.path.join(ROOT, "corb-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    l, r = synth.stereo_pair(4)
    (tmp_path / "l.raw").write_bytes(l.tobytes()); (tmp_path / "r.raw").write_bytes(r.tobytes())
    out = subprocess.check_output([str(exe), str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "1241", "376"]).decode()
    f = dict(kv.split("=") for kv in out.split())
    el, er = pyorc.Extractor(), pyorc.Extractor()
    kl, dl = el.extract(l); kr, dr = er.extract(r); tb = el.tables()
    ur, dp, nm = pyorc.stereo_match(el, er, kl, dl, kr, dr, 386.1448, 718.856, tb["scale"], tb["inv_scale"])
    assert int(f["n_left"]) == len(kl) and int(f["n_right"]) == len(kr) and int(f["matched"]) == nm and f["consistent"] == "1"
    assert int(f["kp_hash"], 16) == _fnv(kl.tobytes()) and int(f["desc_hash"], 16) == _fnv(dl.tobytes())
