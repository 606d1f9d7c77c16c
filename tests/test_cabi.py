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
    assert corb.load().corb_abi_version() == corb.ABI_VERSION == 6      # include/corb_accel.h: CORB_ABI_VERSION (load() refuses a library of another layout)
    import re
    assert int(re.search(r"#define\s+CORB_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "corb_accel.h")).read()).group(1)) == corb.ABI_VERSION
    assert corb.KP_DTYPE.itemsize == 28 and corb.EDGE_DTYPE.itemsize == 24
    assert ctypes.sizeof(corb.OrbConfig) == 36
    assert ctypes.sizeof(corb.BAOptions) == 32            # (scale_factor fills what was padding: the size is part of the C-ABI)


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
                    bad.append(os.path.join(dp, f))
    # comments may cite the oracle as the parity partner, but nothing may load it
    for p in bad:
        txt = open(p, errors="replace").read()
        assert not re.search(r"(dlopen|CDLL|import|#include)[^\n]*(liborc|pyorc|oracle)", txt), p


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    """The C++ host-side mirror of the reference interface (corb-slam_amd/host/corb_host.hpp) builds with
    plain g++ against the C-ABI only (no hip headers, no torch types)."""
    import subprocess
    out = tmp_path / "host_smoke"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "corb-slam_amd", "host"),
           os.path.join(ROOT, "corb-slam_amd", "host", "host_smoke.cpp"), "-o", str(out),
           "-L", os.path.join(ROOT, "corb-slam_amd"), "-lcorb_accel", "-Wl,-rpath," + os.path.join(ROOT, "corb-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    assert out.exists()
    # the signature-preserving adapter templates (ORBmatcher / Optimizer over KeyFrame*, Frame&, Cache*) compile against the test doubles
    out2 = tmp_path / "adapter_main"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "corb-slam_amd", "host"),
                           "-I", os.path.join(ROOT, "tests", "host"), os.path.join(ROOT, "tests", "host", "adapter_main.cpp"), "-o", str(out2),
                           "-L", os.path.join(ROOT, "corb-slam_amd"), "-lcorb_accel", "-Wl,-rpath," + os.path.join(ROOT, "corb-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    assert out2.exists()
    # header is pure C: compiles as C11 too
    c = tmp_path / "t.c"; c.write_text('#include <corb_accel.h>\nint main(void){ CorbKeyPoint k; (void)k; return sizeof(CorbKeyPoint) == 28 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(tmp_path / "t")])
    subprocess.check_call([str(tmp_path / "t")])


def test_bench_cpu_worker_runs_without_a_gpu():
    """bench.py's cpu_baseline throughput sample starts `bench.py --cpu-worker seed n start_at` processes: oracle only, no torch, no GPU"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "liborc_native.so"])
    out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--cpu-worker", "64", "2", "0"], timeout=300).decode().split()
    assert out[-2] == "elapsed" and float(out[-1]) > 0


def test_bench_line_stays_below_eight_kilobytes_and_carries_both_metrics():
    """The driver's record keeps `config`, `roofline`, `cpu_baseline` whole and 8 KB of the line: bench.py's compact form of a full record (the last committed one,
    profiles/r05_bench_k.json -- a 16 KB line) must fit, with metric (ii) -- global-BA LM iterations/s at configs[4]'s size -- inside `roofline` and the same-size
    CPU figure inside `cpu_baseline`."""
    import importlib.util, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    # ... and the full record of round 6's last run (it already carries the two summaries; the informational fixed-tolerance leg and the 1080p counter traffic are in it)
    d6 = json.load(open(os.path.join(root, "profiles", "r06_bench_h_full.json")))
    line6 = json.dumps(b.compact_line(d6), separators=(",", ":"))
    assert len(line6) < 7800 and "ba_config5" in json.loads(line6)["roofline"], len(line6)
    d = json.load(open(os.path.join(root, "profiles", "r05_bench_k.json")))
    d["roofline"]["ba_config5"] = b.ba_summary(d["ba"]["config5"])
    d["cpu_baseline"]["ba_same_size"] = b.ba_cpu_summary(d["ba"]["same_size"])
    c = b.compact_line(d)
    line = json.dumps(c, separators=(",", ":"))
    assert len(line) < 7800, len(line)
    s5 = c["roofline"]["ba_config5"]
    assert s5["poses"] == 50000 and s5["device_iters_per_s"] > 0 and 0 < s5["frac"] < 1 and s5["certificate"]["pcg_residual_max"] < 1e-5
    assert "busy_frac" in s5["schur_mfma"]
    assert c["cpu_baseline"]["ba_same_size"]["kind"] == "port" and c["cpu_baseline"]["ba_same_size"]["value"] > 0
    assert c["config"] == d["config"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in c
