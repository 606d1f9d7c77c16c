"""GPU parity: ORBmatcher::Fuse on records (corb_fuse_store) against the host-pointer form / the oracle on the same scene, and the reference's map update
(AddObservation / AddMapPoint for features without a MapPoint, Replace cases reported) against a sequential restatement of ORBmatcher.cc:1083-1104."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
KF2_ID, KF1_ID = 7, 3


def _scene_on_records(corb, synth, seed, n, span, O=8, exclusive=False):
    rng = np.random.default_rng(seed + 1)
    sc = synth.keyframe_scene(seed, n=n, span=span)
    k2 = sc["kf2"]; nf = len(k2["keys_un"])
    KF = corb.KeyFrameStore(2, nf + 5); MP = corb.MapPointStore(n, O)
    KF.put(1, k2["keys_un"], k2["desc"], k2["u_right"], None, keyframe_id=KF2_ID)
    KF.set_meta(1, id=KF2_ID, client_id=1, flags=0, fx=k2["fx"], fy=k2["fy"], cx=k2["cx"], cy=k2["cy"], bf=k2["bf"], nlevels=8, Tcw=sc["T2w"].astype(np.float32).reshape(16),
                inv_level_sigma2=np.concatenate([k2["inv_level_sigma2"], np.zeros(8, np.float32)]))
    held = np.where(sc["claimed2"] != 0, np.uint64(900000) + np.arange(nf, dtype=np.uint64), NONE)      # features of KF2 that hold a MapPoint already
    KF.set_map_points(1, held)
    pts = sc["pts1"]
    rec = np.zeros(n, corb.MP_RECORD_DTYPE)
    rec["id"] = 1000 + np.arange(n); rec["ref_kf_id"] = KF1_ID; rec["descriptor"] = sc["desc1"]; rec["client_id"] = 1
    rec["world_pos"] = pts["world"]; rec["normal"] = pts["normal"]; rec["min_distance"] = pts["min_distance"]; rec["max_distance"] = pts["max_distance"]
    rec["flags"] = np.where(pts["valid"] != 0, 0, corb.MP_BAD)
    in_kf2 = rng.random(n) < 0.1                                  # already observed by KF2: pMP->IsInKeyFrame(pKF)
    later = rng.random(n) < 0.5                                   # a third observer with a larger id: the new observation goes in the middle of the list
    if exclusive:
        later &= ~in_kf2
    okf, oidx, off = [], [], [0]
    for i in range(n):
        o = [(KF1_ID, i)] + ([(KF2_ID, 0)] if in_kf2[i] else []) + ([(11, 5)] if later[i] else [])
        okf += [a for a, _ in o]; oidx += [b for _, b in o]; off.append(len(okf))
    rec["n_obs"] = np.diff(off)
    MP.put(0, rec, np.array(off, np.int32), np.array(okf, np.uint64), np.array(oidx, np.uint32))
    view = pts.copy(); view["valid"] = ((pts["valid"] != 0) & ~in_kf2).astype(np.uint8)
    cam = corb.TrackCamera.make(k2["fx"], k2["fy"], k2["cx"], k2["cy"], k2["bf"], k2["bf"] / k2["fx"], k2["min_x"], k2["max_x"], k2["min_y"], k2["max_y"], k2["scale"])
    T = sc["T2w"].astype(np.float32)
    Ow = np.array([-(float(T[0, i]) * float(T[0, 3]) + float(T[1, i]) * float(T[1, 3]) + float(T[2, i]) * float(T[2, 3])) for i in range(3)], np.float32)   # -Rcw^T tcw, double accumulation
    return sc, KF, MP, rec, (off, okf, oidx), held, view, cam, T, Ow


@pytest.mark.parametrize("seed,n,span", [(5200, 2000, 1.0), (5201, 2000, 0.3), (5202, 700, 1.0)])
def test_fuse_on_records(corb, pyorc, synth, seed, n, span):
    sc, KF, MP, rec, (off, okf, oidx), held, view, cam, T, Ow = _scene_on_records(corb, synth, seed, n, span)
    k2 = sc["kf2"]; nf = len(k2["keys_un"])
    mt = corb.ORBmatcher(0.6, True)
    for th in (3.0, 4.0):
        h = mt.Fuse(k2, T, Ow, view, sc["desc1"], th)
        r = pyorc.fuse(k2, T, Ow, 0, view, sc["desc1"], th)
        g = KF.Fuse(1, MP, np.arange(n), cam, T, k2["log_scale_factor"], th, apply=False)
        assert np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1]) and g[2] == r[2]
        assert np.array_equal(g[0], h[0]) and np.array_equal(g[1], h[1]) and g[2] == h[2]
        r1, _, _ = MP.get(0, n)
        assert r1.tobytes() == rec.tobytes() and np.array_equal(KF.get_map_points(1), held)         # apply = False: the records are as they were
    assert g[2] > 50
    # the map update, sequentially (ORBmatcher.cc:1083-1104)
    mp = held.copy(); act = np.zeros(n, np.uint8); lists = [list(zip(okf[off[i]: off[i + 1]], oidx[off[i]: off[i + 1]])) for i in range(n)]
    for i in range(n):
        f = int(g[0][i])
        if f < 0:
            continue
        if mp[f] != NONE:
            act[i] = 2
        else:
            act[i] = 1; mp[f] = np.uint64(rec["id"][i]); lists[i] = sorted(lists[i] + [(KF2_ID, f)])
    assert np.array_equal(g[3], act) and (act == 1).sum() > 20 and (act == 2).sum() > 20
    a = KF.Fuse(1, MP, np.arange(n), cam, T, k2["log_scale_factor"], 4.0, apply=True)
    assert np.array_equal(a[0], g[0]) and np.array_equal(a[3], act)
    assert np.array_equal(KF.get_map_points(1), mp)
    r2, k2o, i2o = MP.get(0, n)
    for i in range(n):
        assert r2["n_obs"][i] == len(lists[i]) and [int(x) for x in k2o[i, : len(lists[i])]] == [a_ for a_, _ in lists[i]] and [int(x) for x in i2o[i, : len(lists[i])]] == [b for _, b in lists[i]], i
    b = r2.copy(); b["n_obs"] = rec["n_obs"]
    assert b.tobytes() == rec.tobytes()                                                               # nothing else in the headers moved
    # a second call: the points that entered KF2 are in the keyframe now (IsInKeyFrame) and take no part
    c = KF.Fuse(1, MP, np.arange(n), cam, T, k2["log_scale_factor"], 4.0, apply=True)
    assert (c[0][act == 1] == -1).all() and np.array_equal(c[0][act != 1], g[0][act != 1])
    KF.close(); MP.close()


def test_fuse_on_records_full_observation_list_and_arguments(corb, synth):
    sc, KF, MP, rec, lists, held, view, cam, T, Ow = _scene_on_records(corb, synth, 5210, 600, 1.0, O=2, exclusive=True)      # a point with two observations has no room
    with pytest.raises(corb.CorbError, match="no room"):
        KF.Fuse(1, MP, np.arange(600), cam, T, sc["kf2"]["log_scale_factor"], 4.0, apply=True)
    with pytest.raises(corb.CorbError):
        KF.Fuse(0, MP, np.arange(600), cam, T, sc["kf2"]["log_scale_factor"], 4.0)                            # an empty slot
    with pytest.raises(corb.CorbError):
        KF.Fuse(1, MP, [600], cam, T, sc["kf2"]["log_scale_factor"], 4.0)
    KF.close(); MP.close()
