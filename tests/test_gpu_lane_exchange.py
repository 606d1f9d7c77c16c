"""The cross-lane moves of csrc/lane_exchange.h (DPP instead of ds_bpermute in every wave reduction of the library) against their __shfl_xor / __shfl_up forms, bit for bit,
on the device the tests run on (tools/ubench/lane_exchange.hip: built with the box's hipcc)."""
import os
import shutil
import subprocess
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_exchange_moves_equal_the_shuffle_forms(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "lane_exchange")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "ubench", "lane_exchange.hip")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "DIFFERENT" not in r.stdout and r.stdout.count("equal") == 23, r.stdout + r.stderr
