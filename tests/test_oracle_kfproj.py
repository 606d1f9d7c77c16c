"""Oracle sanity for the keyframe-target matchers (relocalisation SearchByProjection, Fuse x2, SearchBySim3): on a synthetic pair
of keyframes with known correspondences the restated routines must recover (only) true correspondences, and independent
Python loops over the same definitions must agree on the gating arithmetic."""
import numpy as np
import pytest


def _expected_level(max_distance, dist, log_sf, nlevels=8):
    ratio = np.float32(max_distance) / np.float32(dist)
    lg = np.float32(np.log(np.float64(ratio)))
    n = int(np.ceil(np.float32(lg / np.float32(log_sf))))
    return min(max(n, 0), nlevels - 1)


@pytest.mark.parametrize("seed", [5000, 5001])
def test_fuse_recovers_true_features(pyorc, synth, seed):
    sc = synth.keyframe_scene(seed)
    bi, bd, nf = pyorc.fuse(sc["kf2"], sc["T2w"], sc["Ow2"], 0, sc["pts1"], sc["desc1"], 3.0)
    src = sc["true_src2"]
    hits = [i for i in range(len(bi)) if bi[i] >= 0]
    assert nf == len(hits) and nf > 500
    assert all(src[bi[i]] == i for i in hits)                   # descriptors are 5 % noisy copies: only the true feature is within TH_LOW
    assert all(bd[i] <= 50 for i in hits) and all(sc["pts1"]["valid"][i] for i in hits)
    # octave window and chi2 gate of every accepted feature, recomputed independently
    kf = sc["kf2"]; T = sc["T2w"].astype(np.float64)
    for i in hits[:200]:
        X = sc["pts1"]["world"][i].astype(np.float64)
        pc = T[:3, :3] @ X + T[:3, 3]
        dist = np.linalg.norm(X - sc["Ow2"].astype(np.float64))
        lvl = _expected_level(sc["pts1"]["max_distance"][i], np.float32(dist), kf["log_scale_factor"])
        o = kf["keys_un"]["octave"][bi[i]]
        assert lvl - 1 <= o <= lvl
        u = kf["fx"] * pc[0] / pc[2] + kf["cx"]; v = kf["fy"] * pc[1] / pc[2] + kf["cy"]
        e2 = (u - kf["keys_un"]["x"][bi[i]]) ** 2 + (v - kf["keys_un"]["y"][bi[i]]) ** 2
        assert e2 * kf["inv_level_sigma2"][o] < 7.9


def test_fuse_sim3_equals_rigid_for_unit_scale(pyorc, synth):
    """Scw with s = 1 decomposes to the rigid pose: the Sim3 overload finds the rigid overload's features wherever the rigid
    overload's chi2 gate does not bite (it has none), i.e. a superset."""
    sc = synth.keyframe_scene(5002)
    a, _, na = pyorc.fuse(sc["kf2"], sc["T2w"], sc["Ow2"], 0, sc["pts1"], sc["desc1"], 3.0)
    b, _, nb = pyorc.fuse(sc["kf2"], sc["T2w"], None, 1, sc["pts1"], sc["desc1"], 3.0)
    assert nb >= na and all(b[i] == a[i] for i in range(len(a)) if a[i] >= 0)


def test_fuse_invalid_and_behind_camera(pyorc, synth):
    sc = synth.keyframe_scene(5003, n=400)
    pts = sc["pts1"].copy(); pts["valid"] = 0
    bi, bd, nf = pyorc.fuse(sc["kf2"], sc["T2w"], sc["Ow2"], 0, pts, sc["desc1"], 3.0)
    assert nf == 0 and (bi == -1).all()
    pts = sc["pts1"].copy(); pts["world"][:, 2] -= 500.0                       # behind the camera
    bi, bd, nf = pyorc.fuse(sc["kf2"], sc["T2w"], sc["Ow2"], 0, pts, sc["desc1"], 3.0)
    assert nf == 0


@pytest.mark.parametrize("seed", [5010, 5011])
def test_reloc_projection(pyorc, synth, seed):
    sc = synth.keyframe_scene(seed)
    m, n = pyorc.search_by_projection_reloc(sc["kf2"], sc["claimed2"], sc["T2w"], sc["pts1"], sc["desc1"], 10.0, 100, 1)
    src = sc["true_src2"]
    assert n == int((m >= 0).sum()) and n > 400
    assert all(src[f] == m[f] for f in range(len(m)) if m[f] >= 0)
    assert not any(sc["claimed2"][f] for f in range(len(m)) if m[f] >= 0)      # claimed features are never taken
    m2, n2 = pyorc.search_by_projection_reloc(sc["kf2"], sc["claimed2"], sc["T2w"], sc["pts1"], sc["desc1"], 10.0, 100, 0)
    assert n2 >= n                                                             # the rotation histogram only removes matches
    m3, n3 = pyorc.search_by_projection_reloc(sc["kf2"], sc["claimed2"], sc["T2w"], sc["pts1"], sc["desc1"], 10.0, 20, 0)
    assert n3 < n2                                                             # tighter ORBdist


@pytest.mark.parametrize("seed", [5020, 5021])
def test_search_by_sim3_mutual_consistency(pyorc, synth, seed):
    sc = synth.keyframe_scene(seed)
    a = (sc["kf1"], sc["kf2"], sc["T1w"], sc["T2w"], sc["pts1"], sc["desc1"], sc["pts2"], sc["desc2"], sc["s12"], sc["R12"], sc["t12"], 7.5)
    m12, n = pyorc.search_by_sim3(*a)
    src = sc["true_src2"]
    assert n == int((m12 >= 0).sum()) and n > 300
    assert all(src[m12[i]] == i for i in range(len(m12)) if m12[i] >= 0)
    assert len(set(m12[m12 >= 0].tolist())) == n                               # one-to-one
    # already-matched features of KF1 are skipped
    pts1 = sc["pts1"].copy(); pts1["valid"][::2] = 0
    m12b, nb = pyorc.search_by_sim3(sc["kf1"], sc["kf2"], sc["T1w"], sc["T2w"], pts1, sc["desc1"], sc["pts2"], sc["desc2"], sc["s12"], sc["R12"], sc["t12"], 7.5)
    assert (m12b[::2] == -1).all() and nb < n
