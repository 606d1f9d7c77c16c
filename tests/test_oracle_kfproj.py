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


@pytest.mark.parametrize("seed", [5030, 5031])
def test_search_by_projection_scw_recovers_true_features(pyorc, synth, seed):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:425-538): on a scene with known correspondences only true features are taken, features
    held on entry never; with nothing held and uncrowded windows the matches equal Fuse(KeyFrame*, Scw, ...)'s (same gates up to the float / double 1/z) wherever both match."""
    sc = synth.keyframe_scene(seed)
    S = sc["T2w"].copy(); S[:3, :] *= np.float32(1.05)
    m, n = pyorc.search_by_projection_scw(sc["kf2"], sc["claimed2"], S, sc["pts1"], sc["desc1"], 10.0)
    src = sc["true_src2"]
    assert n == int((m >= 0).sum()) and n > 400
    assert all(src[f] == m[f] for f in range(len(m)) if m[f] >= 0)
    assert not any(sc["claimed2"][f] for f in range(len(m)) if m[f] >= 0)
    assert all(sc["pts1"]["valid"][m[f]] for f in range(len(m)) if m[f] >= 0)
    m0, n0 = pyorc.search_by_projection_scw(sc["kf2"], np.zeros(len(m), np.uint8), S, sc["pts1"], sc["desc1"], 10.0)
    bi, bd, nf = pyorc.fuse(sc["kf2"], S, None, 1, sc["pts1"], sc["desc1"], 10.0)
    feat_of_point = np.full(len(bi), -1); feat_of_point[m0[m0 >= 0]] = np.nonzero(m0 >= 0)[0]
    both = (feat_of_point >= 0) & (bi >= 0)
    assert both.sum() > 400 and (feat_of_point[both] == bi[both]).mean() > 0.99


def test_search_by_projection_scw_equals_a_python_loop(pyorc, synth):
    """an independent sequential restatement (numpy float32 arithmetic, brute-force window search in the grid's visiting order replaced by the documented tie rule: the
    first minimum in cell-major order = lowest (cell x, cell y, position in cell)) on a crowded 300-point scene; ties between equal distances are rare but the claims are not"""
    sc = synth.crowd_keyframe_scene(synth.keyframe_scene(5032, n=300, span=0.15), 5032)       # repeated texture: points compete for features
    kf = sc["kf2"]; n = 300
    S = sc["T2w"].copy(); S[:3, :] *= np.float32(0.98)
    claimed0 = sc["claimed2"].copy()
    m, cnt = pyorc.search_by_projection_scw(kf, claimed0, S, sc["pts1"], sc["desc1"], 10.0)
    f32 = np.float32
    scw = f32(np.sqrt(np.float64(S[0, 0]) ** 2 + np.float64(S[0, 1]) ** 2 + np.float64(S[0, 2]) ** 2)); inv = f32(1.0 / np.float64(scw))
    R = (S[:3, :3] * inv).astype(f32); t = (S[:3, 3] * inv).astype(f32)
    Ow = (-(R.T.astype(np.float64)) @ t.astype(np.float64)).astype(f32)
    keys = kf["keys_un"]; claimed = claimed0.copy(); want = np.full(n, -1, np.int32); k = 0
    cellx = np.round((keys["x"] - f32(kf["min_x"])) * f32(64.0 / (kf["max_x"] - kf["min_x"]))).astype(int)       # Frame::PosInGrid rounds (Frame.cc:386-395)
    celly = np.round((keys["y"] - f32(kf["min_y"])) * f32(48.0 / (kf["max_y"] - kf["min_y"]))).astype(int)
    ingrid = (cellx >= 0) & (cellx < 64) & (celly >= 0) & (celly < 48)
    order = [f for f in np.lexsort((np.arange(n), celly, cellx)) if ingrid[f]]       # the grid's visiting order: cell column, cell row, insertion order
    bits = np.unpackbits(kf["desc"], axis=1)
    for i in range(n):
        p = sc["pts1"][i]
        if not p["valid"]:
            continue
        pc = (R.astype(np.float64) @ p["world"].astype(np.float64)).astype(f32) + t if False else ((R.astype(np.float64) @ p["world"].astype(np.float64)) + t.astype(np.float64)).astype(f32)
        if pc[2] < 0:
            continue
        invz = f32(1.0) / pc[2]
        u = f32(kf["fx"]) * (pc[0] * invz) + f32(kf["cx"]); v = f32(kf["fy"]) * (pc[1] * invz) + f32(kf["cy"])
        if not (kf["min_x"] <= u < kf["max_x"] and kf["min_y"] <= v < kf["max_y"]):
            continue
        PO = p["world"] - Ow; dist = f32(np.sqrt((PO.astype(np.float64) ** 2).sum()))
        if dist < f32(0.8) * p["min_distance"] or dist > f32(1.2) * p["max_distance"]:
            continue
        if (PO.astype(np.float64) * p["normal"].astype(np.float64)).sum() < 0.5 * np.float64(dist):
            continue
        lvl = _expected_level(p["max_distance"], dist, kf["log_scale_factor"])
        r = f32(10.0) * kf["scale"][lvl]
        qb = np.unpackbits(sc["desc1"][i])
        best, bidx = 256, -1
        for f in order:
            if not (abs(keys["x"][f] - u) < r and abs(keys["y"][f] - v) < r):
                continue
            if claimed[f] or keys["octave"][f] < lvl - 1 or keys["octave"][f] > lvl:
                continue
            d = int((bits[f] != qb).sum())
            if d < best:
                best, bidx = d, f
        if best <= 50:
            want[bidx] = i; claimed[bidx] = 1; k += 1
    assert np.array_equal(m, want) and cnt == k and k > 30
    # the crowded scene does exercise the claims: without them (every point independently) at least one feature would be chosen twice
    bi, _, _ = pyorc.fuse(kf, S, None, 1, sc["pts1"], sc["desc1"], 10.0)
    taken = bi[bi >= 0]
    assert len(np.unique(taken)) < len(taken)


def test_search_for_initialization_equals_a_python_loop(pyorc, synth):
    """orc_search_for_initialization (ORBmatcher.cc:540-655) against an independent sequential restatement in Python on a crowded 400-feature pair (repeated texture: features
    take matches away from earlier ones), both orientation settings"""
    n = 400
    f1, f2, pm0, _ = synth.monocular_init_pair(7200, n=n, span=0.3, crowd=True, steal_frac=0.2)
    k1, k2 = f1["keys_un"], f2["keys_un"]
    f32 = np.float32
    cellx = np.round((k2["x"] - f32(f2["min_x"])) * f32(64.0 / (f2["max_x"] - f2["min_x"]))).astype(int)
    celly = np.round((k2["y"] - f32(f2["min_y"])) * f32(48.0 / (f2["max_y"] - f2["min_y"]))).astype(int)
    ingrid = (cellx >= 0) & (cellx < 64) & (celly >= 0) & (celly < 48)
    order = [f for f in np.lexsort((np.arange(n), celly, cellx)) if ingrid[f]]
    b1 = np.unpackbits(f1["desc"], axis=1); b2 = np.unpackbits(f2["desc"], axis=1)
    stolen = 0
    for check_ori in (True, False):
        m, pm, cnt = pyorc.search_for_initialization(f1, f2, pm0, 60, 0.9, check_ori)
        INT_MAX = 2 ** 31 - 1
        md = np.full(n, INT_MAX, np.int64); m21 = np.full(n, -1); want = np.full(n, -1, np.int32); k = 0; hist = [[] for _ in range(30)]
        for i1 in range(n):
            if k1["octave"][i1] > 0:
                continue
            x, y = pm0[i1]; r = f32(60)
            best, best2, bidx = INT_MAX, INT_MAX, -1
            for i2 in order:
                if k2["octave"][i2] != 0 or not (abs(k2["x"][i2] - x) < r and abs(k2["y"][i2] - y) < r):
                    continue
                d = int((b1[i1] != b2[i2]).sum())
                if md[i2] <= d:
                    continue
                if d < best: best2, best, bidx = best, d, i2
                elif d < best2: best2 = d
            if best <= 50 and f32(best) < f32(best2) * f32(0.9):
                if m21[bidx] >= 0:
                    want[m21[bidx]] = -1; k -= 1; stolen += 1
                want[i1] = bidx; m21[bidx] = i1; md[bidx] = best; k += 1
                rot = f32(k1["angle"][i1]) - f32(k2["angle"][bidx])
                if rot < 0: rot += f32(360.0)
                b = int(np.round(f32(rot * f32(1.0 / 30)))); b = 0 if b == 30 else b
                hist[b].append(i1)
        if check_ori:
            sizes = [len(h) for h in hist]; o = sorted(range(30), key=lambda i: (-sizes[i], i))
            mx1, mx2, mx3 = sizes[o[0]], sizes[o[1]], sizes[o[2]]
            keep = {o[0]} | ({o[1]} if mx2 >= 0.1 * mx1 else set()) | ({o[2]} if (mx2 >= 0.1 * mx1 and mx3 >= 0.1 * mx1) else set())
            for b in range(30):
                if b in keep: continue
                for i1 in hist[b]:
                    if want[i1] >= 0: want[i1] = -1; k -= 1
        assert np.array_equal(m, want) and cnt == k and k > 40
        mm = want >= 0
        assert np.array_equal(pm[mm], np.stack([k2["x"][want[mm]], k2["y"][want[mm]]], 1)) and np.array_equal(pm[~mm], pm0[~mm])
    assert stolen > 0
