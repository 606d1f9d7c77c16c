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


def test_batch_upload_and_fetch_equal_per_frame_calls(corb, synth):
    """corb_stereo_upload_batch / corb_orb_fetch_batch / corb_stereo_fetch_matches_batch move a whole batch with one copy each and
    return exactly what the per-frame calls return (results strided by corb_orb_capacity)."""
    import numpy as np
    B = 6
    frames = [synth.stereo_pair(40 + i) for i in range(B)]
    sf = corb.StereoFrontend(nfeatures=2000, width=1241, height=376, max_frames=B, fx=718.856, bf=386.1448)
    for s, (l, r) in enumerate(frames):
        sf.upload(s, l, r)
    sf.run(B); sf.sync()
    ref = [sf.fetch(s) for s in range(B)]
    packed = np.ascontiguousarray(np.stack([np.stack([l, r]) for l, r in frames]))
    sf.upload_batch(0, packed); sf.run(B); sf.sync()
    out = sf.fetch_batch(0, B)
    for s in range(B):
        nl, nr = out["counts"][2 * s], out["counts"][2 * s + 1]
        assert np.array_equal(out["kp"][2 * s][:nl], ref[s]["kl"]) and np.array_equal(out["desc"][2 * s][:nl], ref[s]["dl"])
        assert np.array_equal(out["kp"][2 * s + 1][:nr], ref[s]["kr"]) and np.array_equal(out["desc"][2 * s + 1][:nr], ref[s]["dr"])
        assert np.array_equal(out["u_right"][s][:nl].view(np.uint32), ref[s]["u_right"].view(np.uint32)) and out["n_matched"][s] == ref[s]["n_matched"]
    sf.close()


def test_pinned_buffers_round_trip(corb, synth):
    """corb.pinned_empty (hipHostMalloc): batch upload from / fetch into page-locked memory gives the same results as pageable numpy arrays"""
    B = 2
    sf = corb.StereoFrontend(max_frames=B)
    frames = [synth.stereo_pair(20 + i) for i in range(B)]
    packed = np.stack([np.stack(f) for f in frames])
    pin = corb.pinned_empty(packed.shape, np.uint8); pin[...] = packed
    sf.upload_batch(0, pin); sf.run(B); ref = sf.fetch_batch(0, B)
    out = dict((k, corb.pinned_empty(v.shape, v.dtype)) for k, v in ref.items())
    sf.upload_batch(0, packed); sf.run(B); sf.fetch_batch(0, B, out)
    assert np.array_equal(out["counts"], ref["counts"]) and np.array_equal(out["n_matched"], ref["n_matched"])
    for i in range(2 * B):                               # (entries past the counts are not defined)
        m = ref["counts"][i]
        assert np.array_equal(out["kp"][i][:m], ref["kp"][i][:m]) and np.array_equal(out["desc"][i][:m], ref["desc"][i][:m])
    for f in range(B):
        m = ref["counts"][2 * f]
        assert np.array_equal(out["u_right"][f][:m].view(np.uint32), ref["u_right"][f][:m].view(np.uint32)) and np.array_equal(out["depth"][f][:m].view(np.uint32), ref["depth"][f][:m].view(np.uint32))
    sf.close()
