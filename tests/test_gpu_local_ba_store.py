"""GPU parity: Optimizer::LocalBundleAdjustment on store records (corb_local_ba_store) against the oracle's LocalBundleAdjustment on map objects
(oracle/pyorc.py: local_bundle_adjustment): the estimates (1e-4), and -- exactly -- what the reference does to the map afterwards: vToErase
(EraseMapPointMatch / EraseObservation with the reference-keyframe hand-over and SetBadFlag below three observations) and UpdateNormalAndDepth."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFFFFFFFFFF


def _build(corb, synth, seed, n_local=6, n_fixed=4, ppk=25, **kw):
    prob = synth.local_ba_problem(seed=seed, n_local=n_local, n_fixed=n_fixed, pts_per_kf=ppk, **kw)
    K = n_local + n_fixed
    cm = synth.client_maps(prob, 1, K)[0]
    F = max(len(k["kp"]) for k in cm["kf"]) + 3
    KF = corb.KeyFrameStore(K, F); MP = corb.MapPointStore(len(cm["mp_records"]), 16)
    for s, k in enumerate(cm["kf"]):
        KF.put(s, k["kp"], k["desc"], k["ur"], None, keyframe_id=k["id"])
        cam = k["cam"]
        KF.set_meta(s, id=k["id"], client_id=1, flags=0, fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3], bf=cam[4], nlevels=8, Tcw=k["Tcw"].reshape(16),
                    inv_level_sigma2=np.concatenate([k["inv_level_sigma2"], np.zeros(8, np.float32)]))
        KF.set_map_points(s, k["mp_id"])
    MP.put(0, cm["mp_records"], cm["obs_off"], cm["obs_kf"], cm["obs_idx"])
    return prob, cm, KF, MP


def _objects(cm, kf_flags=None, mp_flags=None):
    """the same map as the oracle's objects"""
    kfs = []
    for s, k in enumerate(cm["kf"]):
        fl = 0 if kf_flags is None else kf_flags[s]
        kfs.append(dict(id=int(k["id"]), T=k["Tcw"].reshape(4, 4).copy(), fixed=bool(fl & 2), bad=bool(fl & 1), keys=k["kp"], ur=k["ur"], mp=[int(x) for x in k["mp_id"]],
                        intr=[float(c) for c in k["cam"]], nlevels=8, inv_level_sigma2=k["inv_level_sigma2"]))
    by = {k["id"]: k for k in kfs}
    mps = []
    rec, off = cm["mp_records"], cm["obs_off"]
    for j in range(len(rec)):
        obs = {int(cm["obs_kf"][t]): int(cm["obs_idx"][t]) for t in range(off[j], off[j + 1])}
        fl = int(rec["flags"][j]) if mp_flags is None else mp_flags[j]
        mps.append(dict(id=int(rec["id"][j]), pos=rec["world_pos"][j].copy(), fixed=bool(fl & 2), bad=bool(fl & 1), obs=obs, ref=int(rec["ref_kf_id"][j]),
                        nObs=sum(2 if by[kid]["ur"][f] >= 0 else 1 for kid, f in obs.items()), normal=rec["normal"][j].copy(),
                        min_distance=rec["min_distance"][j], max_distance=rec["max_distance"][j]))
    return kfs, mps


def _compare_map(corb, KF, MP, kfs, mps, n_local, tol=1e-4):
    rec, okf, oidx = MP.get(0, len(mps))
    n_bad = 0
    for j, m in enumerate(mps):
        assert bool(rec["flags"][j] & corb.MP_BAD) == m["bad"], j
        assert rec["n_obs"][j] == len(m["obs"]), j
        ids = sorted(m["obs"])
        assert [int(x) for x in okf[j, : len(ids)]] == ids and [int(x) for x in oidx[j, : len(ids)]] == [m["obs"][i] for i in ids], j
        n_bad += m["bad"]
        if not m["bad"]:
            assert rec["ref_kf_id"][j] == m["ref"], j
        assert np.abs(rec["world_pos"][j] - m["pos"]).max() <= tol * max(1.0, np.abs(m["pos"]).max()), j
        assert np.abs(rec["normal"][j] - m["normal"]).max() <= tol, j
        assert abs(rec["max_distance"][j] - m["max_distance"]) <= tol * max(1.0, m["max_distance"]) and abs(rec["min_distance"][j] - m["min_distance"]) <= tol * max(1.0, m["min_distance"]), j
    for s, k in enumerate(kfs):
        got = KF.get_map_points(s)
        want = np.array([NONE if q is None else q for q in k["mp"]], np.uint64)
        assert np.array_equal(got, want), s
        T = KF.get_meta(s)["Tcw"].reshape(4, 4)
        if s >= n_local:
            assert np.array_equal(T, k["T"])                               # lFixedCameras are not written
        else:
            assert np.abs(T - k["T"]).max() <= tol * max(1.0, np.abs(k["T"]).max()), s
    return n_bad


def _device_route(g):
    """corb_local_ba_store flattened the window on the device (corb_ba_staged_device): that route solves on the FULL block pattern, the host route on the exact one
    -- and a window small enough for the one-workgroup optimiser (BA_SMALL_EDGES observations) is declined by it"""
    st = g["structure"]
    assert g["device_route"] == (st["nnz_blocks"] == st["free_poses"] ** 2 and st["schur_pairs"] > 0)
    return g["device_route"]


HOST_ROUTE = os.environ.get("CORB_LBA_HOST_FLATTEN") is not None      # (development: the library was told to take the host route everywhere)


@pytest.mark.parametrize("seed,ppk", [(2100, 25), (2101, 25), (2102, 25), (2103, 140), (2104, 140)])
def test_local_ba_on_records_matches_the_oracle_on_map_objects(corb, pyorc, synth, seed, ppk):
    """ppk = 25: a window for the one-workgroup optimiser (host route); 140: ~3 000 observations -- flattened, optimised and classified on the device"""
    n_local = 6
    prob, cm, KF, MP = _build(corb, synth, seed, n_local=n_local, ppk=ppk, outlier_frac=0.12, max_obs=5)
    kfs, mps = _objects(cm)
    K, M = len(kfs), len(mps)
    o = pyorc.local_bundle_adjustment(kfs[:n_local], kfs[n_local:], mps, scale_factor=1.2)
    g = corb.LocalBundleAdjustmentStore(KF, np.arange(K), n_local, MP, np.arange(M), scale_factor=1.2)
    assert HOST_ROUTE or _device_route(g) == (ppk > 100), g["structure"]
    assert len(g["erase"]) > 0 and sorted(map(tuple, g["erase"].tolist())) == sorted(o["erase"])
    assert np.abs(g["poses"] - o["poses"]).max() <= 1e-4 * max(1.0, np.abs(o["poses"]).max()) and np.abs(g["points"] - o["points"]).max() <= 1e-4 * max(1.0, np.abs(o["points"]).max())
    n_bad = _compare_map(corb, KF, MP, kfs, mps, n_local)
    assert n_bad > 0                                                       # SetBadFlag happened
    moved = [j for j, m in enumerate(mps) if not m["bad"] and m["ref"] != int(cm["mp_records"]["ref_kf_id"][j])]
    assert len(moved) > 0                                                  # ... and so did the hand-over of mpRefKF
    # UpdateNormalAndDepth from the PRODUCT's own estimates: the arithmetic alone, to float rounding
    rec, _, _ = MP.get(0, M)
    for s, k in enumerate(kfs):
        k["T"] = KF.get_meta(s)["Tcw"].reshape(4, 4).copy()
    by = {k["id"]: k for k in kfs}
    for j, m in enumerate(mps):
        if m["bad"] or m["fixed"]:
            continue
        m["pos"] = rec["world_pos"][j].copy()
        pyorc.map_point_update_normal_and_depth(m, by, 1.2)
        assert np.abs(rec["normal"][j] - m["normal"]).max() <= 2e-7 and abs(rec["max_distance"][j] - m["max_distance"]) <= 2e-6 * m["max_distance"] and abs(rec["min_distance"][j] - m["min_distance"]) <= 2e-6 * m["min_distance"], j
    KF.close(); MP.close()


def test_local_ba_on_records_forty_local_keyframes_on_the_device_route(corb, pyorc, synth):
    """40 local keyframes: the device route with the blocked dense solve (the reduced system is 234 x 234: no one-workgroup solve, no chains of LM iterations) on the
    full 39 x 39 block pattern"""
    n_local = 40
    prob, cm, KF, MP = _build(corb, synth, 2140, n_local=n_local, n_fixed=3, ppk=20, outlier_frac=0.1, max_obs=5)
    kfs, mps = _objects(cm)
    K, M = len(kfs), len(mps)
    o = pyorc.local_bundle_adjustment(kfs[:n_local], kfs[n_local:], mps, scale_factor=1.2)
    g = corb.LocalBundleAdjustmentStore(KF, np.arange(K), n_local, MP, np.arange(M), scale_factor=1.2)
    assert HOST_ROUTE or (_device_route(g) and 32 < g["structure"]["free_poses"] <= 64), g["structure"]
    assert sorted(map(tuple, g["erase"].tolist())) == sorted(o["erase"]) and len(o["erase"]) > 0
    assert np.abs(g["poses"] - o["poses"]).max() <= 1e-4 * max(1.0, np.abs(o["poses"]).max()) and np.abs(g["points"] - o["points"]).max() <= 1e-4 * max(1.0, np.abs(o["points"]).max())
    _compare_map(corb, KF, MP, kfs, mps, n_local)
    KF.close(); MP.close()


def test_local_ba_on_records_window_the_device_route_declines(corb, pyorc, synth):
    """70 local keyframes: more free keyframes than the device route's full block pattern takes (64) -- corb_ba_staged_device declines after its counts, and the host
    route goes on with the graph that is already on the device (its edge array sized by the bound, the edge count from the declined call)"""
    n_local = 70
    prob, cm, KF, MP = _build(corb, synth, 2130, n_local=n_local, n_fixed=2, ppk=10, outlier_frac=0.1, max_obs=5)
    kfs, mps = _objects(cm)
    K, M = len(kfs), len(mps)
    o = pyorc.local_bundle_adjustment(kfs[:n_local], kfs[n_local:], mps, scale_factor=1.2)
    g = corb.LocalBundleAdjustmentStore(KF, np.arange(K), n_local, MP, np.arange(M), scale_factor=1.2)
    assert not g["device_route"] and g["structure"]["free_poses"] > 64
    assert sorted(map(tuple, g["erase"].tolist())) == sorted(o["erase"]) and len(o["erase"]) > 0
    assert np.abs(g["poses"] - o["poses"]).max() <= 1e-4 * max(1.0, np.abs(o["poses"]).max()) and np.abs(g["points"] - o["points"]).max() <= 1e-4 * max(1.0, np.abs(o["points"]).max())
    _compare_map(corb, KF, MP, kfs, mps, n_local)
    KF.close(); MP.close()


@pytest.mark.parametrize("ppk,fixed_seen_by_fixed", [(25, True), (160, True), (160, False)])
def test_local_ba_on_records_flags_and_no_erase(corb, pyorc, synth, ppk, fixed_seen_by_fixed):
    """CORB_KF_FIXED among the local keyframes, a fixed and a bad map point, a bad keyframe among the fixed ones; apply_erase = 0 leaves the lists alone.
    fixed_seen_by_fixed: the fixed point is (also) observed by keyframes that do not move -- an edge between a fixed keyframe and a fixed point is outside the flattened
    graph (g2o never activates it); the device route classifies it by its constant depth (flat_launch_fixed_edge_outliers)"""
    n_local = 5
    prob, cm, KF, MP = _build(corb, synth, 2110, n_local=n_local, n_fixed=4, ppk=ppk, outlier_frac=0.1, max_obs=6)
    K, M = len(cm["kf"]), len(cm["mp_records"])
    kf_flags = [0] * K; kf_flags[2] = 2; kf_flags[7] = 1
    still = {int(cm["kf"][s]["id"]) for s in range(K) if s >= n_local or s == 2}          # keyframes that do not move: the fixed cameras and the getFixed() one
    off = cm["obs_off"]
    seen_by_still = [any(int(k) in still for k in cm["obs_kf"][off[j]:off[j + 1]]) for j in range(M)]
    fixed_pt = next(j for j in range(M) if j != 11 and off[j + 1] - off[j] >= 2 and seen_by_still[j] == fixed_seen_by_fixed)
    mp_flags = [int(x) for x in cm["mp_records"]["flags"]]; mp_flags[fixed_pt] |= 2; mp_flags[11] |= 1
    for s in (2, 7):
        KF.set_meta(s, flags=kf_flags[s])
    r, okf, oidx = MP.get(0, M); r["flags"] = mp_flags
    MP.put(0, r, cm["obs_off"], cm["obs_kf"], cm["obs_idx"])
    for erase in (False, True):
        kfs, mps = _objects(cm, kf_flags, mp_flags)
        if erase:                                                          # second round: starts from the first round's records
            for s, k in enumerate(kfs):
                k["T"] = KF.get_meta(s)["Tcw"].reshape(4, 4).copy()
            r1, _, _ = MP.get(0, M)
            for j, m in enumerate(mps):
                m["pos"] = r1["world_pos"][j].copy(); m["normal"] = r1["normal"][j].copy(); m["min_distance"] = r1["min_distance"][j]; m["max_distance"] = r1["max_distance"][j]
        before = KF.get_meta(7).tobytes()
        o = pyorc.local_bundle_adjustment(kfs[:n_local], kfs[n_local:], mps, scale_factor=1.2, apply_erase=erase)
        g = corb.LocalBundleAdjustmentStore(KF, np.arange(K), n_local, MP, np.arange(M), scale_factor=1.2, apply_erase=erase)
        assert HOST_ROUTE or _device_route(g) == (ppk > 100), g["structure"]
        assert sorted(map(tuple, g["erase"].tolist())) == sorted(o["erase"]) and len(o["erase"]) > 0
        assert not any(p == 7 for p, _ in o["erase"]) and not any(j == 11 for _, j in o["erase"])      # a bad keyframe / a bad point has no edges
        _compare_map(corb, KF, MP, kfs, mps, n_local)
        assert KF.get_meta(7).tobytes() == before
        assert np.array_equal(KF.get_meta(2)["Tcw"], cm["kf"][2]["Tcw"].reshape(16))                    # getFixed(): not written
        rr, _, _ = MP.get(0, M)
        assert np.array_equal(rr["world_pos"][fixed_pt], cm["mp_records"]["world_pos"][fixed_pt]) and np.array_equal(rr["world_pos"][11], cm["mp_records"]["world_pos"][11])
        if not erase:
            assert np.array_equal(rr["n_obs"], cm["mp_records"]["n_obs"])
    KF.close(); MP.close()


def test_local_ba_on_records_stop_flag_and_errors(corb, synth):
    prob, cm, KF, MP = _build(corb, synth, 2120)
    K, M = len(cm["kf"]), len(cm["mp_records"])
    r0, _, _ = MP.get(0, M); m0 = [KF.get_meta(s).tobytes() for s in range(K)]
    g = corb.LocalBundleAdjustmentStore(KF, np.arange(K), 6, MP, np.arange(M), stop="before")          # if(*pbStopFlag) return; nothing is touched
    r1, _, _ = MP.get(0, M)
    assert len(g["erase"]) == 0 and r1.tobytes() == r0.tobytes() and [KF.get_meta(s).tobytes() for s in range(K)] == m0
    with pytest.raises(corb.CorbError):
        corb.LocalBundleAdjustmentStore(KF, np.arange(K), K + 1, MP, np.arange(M))
    with pytest.raises(corb.CorbError):
        corb.LocalBundleAdjustmentStore(KF, [0, 1, 1], 2, MP, np.arange(M))                            # a keyframe twice
    with pytest.raises(corb.CorbError, match="named twice"):
        corb.LocalBundleAdjustmentStore(KF, np.arange(K), 6, MP, np.r_[np.arange(M), 3])               # a map point twice: two vertices, one record
    with pytest.raises(corb.CorbError, match="named twice"):
        corb.GlobalBundleAdjustemntStore(KF, np.arange(K), MP, np.r_[5, np.arange(M)], nIterations=2)
    KF.close(); MP.close()


def test_global_ba_on_records_update_normal_and_depth(corb, pyorc, synth):
    """corb_ba_solve_store, nLoopKF == 0: with options.scale_factor > 0 SetWorldPos is followed by UpdateNormalAndDepth on the records (Optimizer.cc:254-256,
    MapPoint.cc:424-472); with 0 the three fields stay as they were filed."""
    prob, cm, KF, MP = _build(corb, synth, 77, n_local=6, n_fixed=4, ppk=25)
    K, M = len(cm["kf"]), len(cm["mp_records"])
    before, _, _ = MP.get(0, M)
    corb.GlobalBundleAdjustemntStore(KF, list(range(K)), MP, list(range(M)), nIterations=3, nLoopKF=0)
    rec0, _, _ = MP.get(0, M)
    assert np.array_equal(rec0["normal"], before["normal"]) and np.array_equal(rec0["max_distance"], before["max_distance"]) and np.array_equal(rec0["min_distance"], before["min_distance"])
    assert np.abs(rec0["world_pos"] - before["world_pos"]).max() > 0
    corb.GlobalBundleAdjustemntStore(KF, list(range(K)), MP, list(range(M)), nIterations=3, nLoopKF=0, scale_factor=1.2)
    rec, _, _ = MP.get(0, M)
    kfs, mps = _objects(cm)
    for s, k in enumerate(kfs):
        k["T"] = KF.get_meta(s)["Tcw"].reshape(4, 4).copy()
    by = {k["id"]: k for k in kfs}
    n_checked = 0
    for j, m in enumerate(mps):
        if m["bad"] or m["fixed"]:
            continue
        m["pos"] = rec["world_pos"][j].copy()
        pyorc.map_point_update_normal_and_depth(m, by, 1.2)
        assert np.abs(rec["normal"][j] - m["normal"]).max() <= 2e-7 and abs(rec["max_distance"][j] - m["max_distance"]) <= 2e-6 * m["max_distance"] and abs(rec["min_distance"][j] - m["min_distance"]) <= 2e-6 * m["min_distance"], j
        n_checked += 1
    assert n_checked > 50 and np.abs(rec["normal"] - before["normal"]).max() > 0
    # nLoopKF != 0 writes TcwGBA / pos_gba only: the fields stay
    corb.GlobalBundleAdjustemntStore(KF, list(range(K)), MP, list(range(M)), nIterations=2, nLoopKF=5, scale_factor=1.2)
    rec2, _, _ = MP.get(0, M)
    assert np.array_equal(rec2["normal"], rec["normal"]) and np.array_equal(rec2["max_distance"], rec["max_distance"])
    KF.close(); MP.close()
