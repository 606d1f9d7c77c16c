"""GPU tests: the tracking-thread calls on device-resident records (corb_track_search_last_frame, corb_track_pose_optimization) against the host-pointer calls
of the same operations (which tests/test_gpu_proj.py, test_gpu_staged.py and test_gpu_replay.py hold against the oracle): identical matches, poses,
outlier sets, and the records updated as Tracking::TrackWithMotionModel would update the Frame (C/src/Tracking.cc:868-940)."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


def _put_points(mp, rec):
    """records with rec["n_obs"] observations each (the store takes Observations() from the observation lists)"""
    off = np.concatenate([[0], np.cumsum(rec["n_obs"])]).astype(np.int32)
    mp.put(0, rec, off, np.ones(off[-1], np.uint64), np.zeros(off[-1], np.uint32))


def _scene(corb, seed=77, t_last=5, t_cur=6, bad_every=17, unobserved_every=23, outlier_every=29, drop_every=5):
    import replay_client as rc
    w = rc.World(seed, 12)
    CAM = rc.CAM
    f32 = lambda x: float(np.float32(x))
    cam = corb.TrackCamera.make(f32(CAM["fx"]), f32(CAM["fy"]), f32(CAM["cx"]), f32(CAM["cy"]), f32(CAM["bf"]), f32(np.float32(CAM["bf"]) / np.float32(CAM["fx"])),
                                0.0, float(CAM["w"]), 0.0, float(CAM["h"]), w.scale)
    nL = len(w.X)
    ids = (np.arange(nL, dtype=np.uint64) * np.uint64(3) + np.uint64(1000))          # ids are not slots
    rec = np.zeros(nL, corb.MP_RECORD_DTYPE)
    rec["id"] = ids; rec["world_pos"] = w.Xest; rec["descriptor"] = w.desc; rec["n_obs"] = 1
    rec["flags"][::bad_every] = 1                                                    # isBad()
    rec["n_obs"][::unobserved_every] = 0                                             # Observations() == 0: matches but does not claim
    mp = corb.MapPointStore(nL, 4)
    _put_points(mp, rec)
    mp.build_index(0, nL)
    kf = corb.KeyFrameStore(4, 2048)
    inv_s2 = np.zeros(16, np.float32); inv_s2[:8] = (1.0 / (w.scale * w.scale)).astype(np.float32)
    frames = {}
    for slot, t in ((0, t_last), (1, t_cur)):
        keys, ur, desc, lm = w.observe(t)
        kf.put(slot, keys, desc, ur, None, keyframe_id=slot + 1)
        kf.set_meta(slot, id=slot + 1, fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, bf=cam.bf, nlevels=8, inv_level_sigma2=inv_s2, Tcw=w.pose(t).astype(np.float32))
        frames[slot] = dict(keys=keys, ur=ur, desc=desc, lm=lm, T=w.pose(t).astype(np.float32))
    last = frames[0]
    has = np.ones(len(last["lm"]), bool); has[::drop_every] = False                  # features of the last frame without a MapPoint
    outl = np.zeros(len(last["lm"]), bool); outl[::outlier_every] = True             # mvbOutlier of the last frame
    kf.set_map_points(0, np.where(has, ids[last["lm"]], NONE))
    kf.set_flags(0, np.where(outl, 2, 0).astype(np.uint8))
    last.update(has=has, outl=outl)
    return w, cam, ids, rec, mp, kf, frames


def test_search_last_frame_and_pose_optimization_on_records(corb):
    import replay_client as rc
    w, cam, ids, rec, mp, kf, fr = _scene(corb)
    last, cur = fr[0], fr[1]
    T_pred = (cur["T"].astype(np.float64) @ np.linalg.inv(np.eye(4))).astype(np.float32)
    T_pred[0, 3] += 0.02                                                             # a motion-model prediction, slightly off
    # ---- the host-pointer call on the same inputs ----
    lm = last["lm"]
    lastp = np.zeros(len(lm), corb.LAST_DTYPE)
    lastp["world"] = w.Xest[lm]; lastp["angle"] = last["keys"]["angle"]; lastp["octave"] = last["keys"]["octave"]
    usable = last["has"] & ~last["outl"] & (rec["flags"][lm] == 0)
    lastp["valid"] = usable; lastp["claims"] = rec["n_obs"][lm] > 0
    ldesc = np.where(usable[:, None], w.desc[lm], 0).astype(np.uint8)
    fv = rc._frame_view(w, cur["keys"], cur["ur"], cur["desc"])
    matcher = corb.ORBmatcher(0.9, True)
    m_ref, n_ref = matcher.SearchByProjection_Frame(fv, T_pred, last["T"], cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, cam.mb, lastp, ldesc, 7.0, False)
    assert n_ref > 800
    # ---- on records ----
    m, n = kf.TrackSearchLastFrame(1, 0, mp, T_pred, last["T"], cam, 7.0, mono=False)
    assert n == n_ref and np.array_equal(m, m_ref)
    got = kf.get_map_points(1)
    want = np.where(m_ref >= 0, ids[lm[np.maximum(m_ref, 0)]], NONE)
    assert np.array_equal(got, want)                                                 # CurrentFrame.mvpMapPoints
    assert np.array_equal(kf.get_map_points(0), np.where(last["has"], ids[lm], NONE))   # the last frame is untouched
    # ---- PoseOptimization(&CurrentFrame) ----
    sel = np.nonzero(m_ref >= 0)[0]                                                  # features in index order, as the reference adds its edges
    f_lm = lm[m_ref[sel]]
    pts = w.Xest[f_lm]; obs = np.stack([cur["keys"]["x"][sel], cur["keys"]["y"][sel], cur["ur"][sel]], 1).astype(np.float32)
    wgt = (1.0 / (w.scale[cur["keys"]["octave"][sel]] ** 2)).astype(np.float32)
    T_ref, out_ref, inl_ref = corb.Optimizer.PoseOptimizationBatch([(T_pred, pts, obs, wgt)], cam.fx, cam.fy, cam.cx, cam.cy, cam.bf)[0]
    T, outl, inl = kf.TrackPoseOptimization(1, mp, cam, T_pred)
    assert np.array_equal(np.asarray(T).reshape(16), np.asarray(T_ref).reshape(16))
    full = np.zeros(len(cur["keys"]), bool); full[sel] = np.asarray(out_ref, bool)
    assert np.array_equal(outl, full) and inl == inl_ref and 0 < full.sum() < len(sel)
    assert np.array_equal(np.asarray(kf.get_meta(1)["Tcw"]).reshape(16), np.asarray(T).reshape(16))      # pFrame->SetPose
    assert np.abs(np.asarray(T).reshape(4, 4)[:3, 3] - w.pose(6)[:3, 3]).max() < 0.05
    # the outlier flags are what the NEXT frame's search skips: make frame 1 the last frame of frame 2
    keys, ur, desc, lm2 = w.observe(7)
    kf.put(2, keys, desc, ur, None, keyframe_id=3)
    inv_s2 = np.zeros(16, np.float32); inv_s2[:8] = (1.0 / (w.scale * w.scale)).astype(np.float32)
    kf.set_meta(2, id=3, nlevels=8, inv_level_sigma2=inv_s2)
    m2, n2 = kf.TrackSearchLastFrame(2, 1, mp, w.pose(7).astype(np.float32), T, cam, 7.0)
    lm1 = np.where(m_ref >= 0, lm[np.maximum(m_ref, 0)], 0)
    lastp2 = np.zeros(len(cur["keys"]), corb.LAST_DTYPE)
    use2 = (m_ref >= 0) & ~full
    lastp2["world"] = w.Xest[lm1]; lastp2["angle"] = cur["keys"]["angle"]; lastp2["octave"] = cur["keys"]["octave"]; lastp2["valid"] = use2; lastp2["claims"] = rec["n_obs"][lm1] > 0
    m2_ref, n2_ref = matcher.SearchByProjection_Frame(rc._frame_view(w, keys, ur, desc), w.pose(7).astype(np.float32), T, cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, cam.mb,
                                                      lastp2, np.where(use2[:, None], w.desc[lm1], 0).astype(np.uint8), 7.0, False)
    assert n2 == n2_ref and np.array_equal(m2, m2_ref) and n2 > 500
    # "Discard outliers" (Tracking.cc:919-940): the rejected features lose their MapPoint, keep no outlier flag, and the next stages do not see them
    ids2_before = kf.get_map_points(2)
    T2, out2, inl2 = kf.TrackPoseOptimization(2, mp, cam, w.pose(7).astype(np.float32), discard_outliers=True)
    fl2 = kf.get(2)["flags"]
    assert out2.sum() > 0 and np.array_equal((fl2 & 4) != 0, out2) and not (fl2 & 2).any()
    assert np.array_equal(kf.get_map_points(2), ids2_before)                          # (the ids stay in the record as mnLastFrameSeen)
    T2b, out2b, inl2b = kf.TrackPoseOptimization(2, mp, cam, T2)                        # the discarded features carry no edge any more
    assert inl2b + out2b.sum() == int((m2_ref >= 0).sum()) - int(out2.sum()) and not (out2b & out2).any()
    kf.close(); mp.close()


def test_track_calls_edge_cases(corb):
    w, cam, ids, rec, mp, kf, fr = _scene(corb, seed=78)
    cur = fr[1]
    T0 = cur["T"]
    # no map points in the frame: `if(nInitialCorrespondences<3) return 0;` -- pose passed through, nothing flagged
    T, outl, inl = kf.TrackPoseOptimization(1, mp, cam, T0)
    assert np.array_equal(np.asarray(T).reshape(16), T0.reshape(16)) and inl == 0 and not outl.any()
    # two map points only
    two = np.full(len(cur["keys"]), NONE, np.uint64); two[3] = ids[cur["lm"][3]]; two[9] = ids[cur["lm"][9]]
    kf.set_map_points(1, two)
    T, outl, inl = kf.TrackPoseOptimization(1, mp, cam, T0)
    assert np.array_equal(np.asarray(T).reshape(16), T0.reshape(16)) and inl == 0 and not outl.any()
    # ids that are not in the store are no map points
    kf.set_map_points(1, np.full(len(cur["keys"]), np.uint64(7), np.uint64))
    T, outl, inl = kf.TrackPoseOptimization(1, mp, cam, T0)
    assert inl == 0
    # an empty local list: nothing to match, but the frame's bad points still leave it (SearchLocalPoints' first loop)
    lm = cur["lm"]; bad_l = np.nonzero(rec["flags"][lm] != 0)[0]
    assert len(bad_l) > 3
    kf.set_map_points(1, ids[lm])
    m, n, nv = kf.TrackSearchLocalPoints(1, mp, np.zeros(0, np.uint64), cam, T0, float(np.float32(np.log(np.float32(1.2)))))
    assert n == 0 and nv == 0 and (m == -1).all()
    after = kf.get_map_points(1)
    assert (after[bad_l] == NONE).all() and np.array_equal(np.delete(after, bad_l), np.delete(ids[lm], bad_l))
    # a last frame without usable points: no matches, nothing written
    kf.set_map_points(0, np.full(len(fr[0]["lm"]), NONE, np.uint64)); kf.set_map_points(1, np.full(len(lm), NONE, np.uint64))
    m, n = kf.TrackSearchLastFrame(1, 0, mp, T0, fr[0]["T"], cam, 7.0)
    assert n == 0 and (m == -1).all() and (kf.get_map_points(1) == NONE).all()
    # an empty slot, equal slots, a store without index
    with pytest.raises(RuntimeError):
        kf.TrackSearchLastFrame(3, 0, mp, T0, T0, cam, 7.0)
    with pytest.raises(RuntimeError):
        kf.TrackSearchLastFrame(1, 1, mp, T0, T0, cam, 7.0)
    mp2 = corb.MapPointStore(8, 2)
    with pytest.raises(RuntimeError):
        kf.TrackPoseOptimization(1, mp2, cam, T0)
    # a put makes the index stale: the calls refuse until it is rebuilt
    _put_points(mp, rec)
    with pytest.raises(RuntimeError):
        kf.TrackPoseOptimization(1, mp, cam, T0)
    mp.build_index(0, len(rec)); kf.TrackPoseOptimization(1, mp, cam, T0)
    # duplicate ids in the indexed range
    r2 = np.zeros(8, corb.MP_RECORD_DTYPE); r2["id"] = 5
    mp2.put(0, r2, np.zeros(9, np.int32), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    with pytest.raises(RuntimeError):
        mp2.build_index(0, 8)
    kf.close(); mp.close(); mp2.close()


def test_search_local_points_on_records(corb, pyorc):
    """Tracking::SearchLocalPoints: isInFrustum on the device against the numpy restatement (bit-equal TRACKED values), the matches against the host-pointer
    SearchByProjection(Frame&, vpMapPoints, th) on those values (itself held against the oracle in tests/test_gpu_proj.py and the replay)"""
    import replay_client as rc
    w, cam, ids, rec, mp, kf, fr = _scene(corb, seed=79)
    cur = fr[1]; lm = cur["lm"]
    # normals / distance ranges of the map points as MapPoint::UpdateNormalAndDepth would leave them for an observer near the trajectory
    nL = len(w.X)
    C0 = np.array([0.0, 0.0, 4.0], np.float32)
    PO = w.Xest - C0; dist = np.linalg.norm(PO, axis=1).astype(np.float32)
    rec["normal"] = (PO / dist[:, None]).astype(np.float32)
    rec["max_distance"] = (dist * w.scale[w.octave] * np.float32(1.3)).astype(np.float32); rec["min_distance"] = (rec["max_distance"] / w.scale[7] / np.float32(1.5)).astype(np.float32)
    _put_points(mp, rec); mp.build_index(0, nL)
    # the frame already holds every third of its landmarks (TrackWithMotionModel's matches), one of them a bad point
    held = np.zeros(len(lm), bool); held[::3] = True
    kf.set_map_points(1, np.where(held, ids[lm], NONE))
    disc = np.zeros(len(lm), bool); disc[::12] = True; disc &= held & (rec["flags"][lm] == 0)     # held points that an earlier PoseOptimization discarded as outliers:
    kf.set_flags(1, np.where(disc, 4, 0).astype(np.uint8))                                        # no MapPoint any more, but seen in this frame (mnLastFrameSeen)
    T = cur["T"]
    # mvpLocalMapPoints: the landmarks of the last / current frame plus strangers (behind the camera, out of range, unknown ids), shuffled
    rng = np.random.default_rng(5)
    local = np.unique(np.concatenate([fr[0]["lm"], lm, rng.integers(0, nL, 1500)]))
    rng.shuffle(local)
    local_ids = np.concatenate([ids[local], np.array([5, 6, 7], np.uint64)])               # three ids the store does not know
    logs = float(np.float32(np.log(np.float32(1.2))))
    m, n, inview, tr = kf.TrackSearchLocalPoints(1, mp, local_ids, cam, T, logs, th=1.0, nnratio=0.8, want_tracked=True)
    # ---- expected: isInFrustum of the candidates that are known, not bad and not in the frame ----
    in_frame = np.isin(local, lm[held & (rec["flags"][lm] == 0)])
    cand = (rec["flags"][local] == 0) & ~in_frame
    exp = pyorc.is_in_frustum(T, w.Xest[local], rec["normal"][local], rec["min_distance"][local], rec["max_distance"][local], cam.fx, cam.fy, cam.cx, cam.cy, cam.bf,
                              0.0, float(rc.CAM["w"]), 0.0, float(rc.CAM["h"]), logs, 8)
    exp["valid"] &= cand; exp["claims"] = (rec["n_obs"][local] > 0) & exp["valid"].astype(bool)
    for k in ("proj_x", "proj_y", "proj_xr", "view_cos", "level"):
        exp[k] = np.where(exp["valid"], exp[k], 0)
    trk = tr[: len(local)]
    assert tr[len(local):]["valid"].sum() == 0
    assert np.array_equal(trk["valid"], exp["valid"]) and inview == int(exp["valid"].sum()) and 300 < inview < len(local)
    for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
        assert np.array_equal(trk[k].view(np.uint32), exp[k].view(np.uint32)), k
    assert np.array_equal(trk["level"], exp["level"]) and np.array_equal(trk["claims"], exp["claims"])
    # ---- the matcher on those values ----
    claimed = (held & ~disc & (rec["flags"][lm] == 0) & (rec["n_obs"][lm] > 0)).astype(np.uint8)
    fv = rc._frame_view(w, cur["keys"], cur["ur"], cur["desc"], claimed=claimed)
    m_ref, n_ref = corb.ORBmatcher(0.8, True).SearchByProjection(fv, exp, np.where(exp["valid"][:, None].astype(bool), w.desc[local], 0).astype(np.uint8), 1.0)
    assert n == n_ref and np.array_equal(m, m_ref) and n > 100
    got = kf.get_map_points(1)
    before = np.where(held & (rec["flags"][lm] == 0), ids[lm], NONE)                        # the bad points left the frame
    want = np.where(m_ref >= 0, local_ids[np.maximum(m_ref, 0)], before)
    assert np.array_equal(got, want)
    kf.close(); mp.close()
