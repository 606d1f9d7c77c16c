"""GPU parity of the keyframe-target matchers ON RECORDS (VERDICT r4 missing 5; r5 missing 1: corb_search_by_projection_scw_store): corb_track_search_reloc (ORBmatcher::SearchByProjection(Frame&, KeyFrame*,
sAlreadyFound, th, ORBdist), C/src/ORBmatcher.cc:1616-1744) and corb_search_by_sim3_store (SearchBySim3, :1244-1468) against the oracle on the flat views of the same
scene, with the pointer-level tests (NULL / isBad() / sAlreadyFound / vbAlreadyMatched through GetIndexInKeyFrame) evaluated from the records on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


def _stores(corb, sc, n, ids1, ids2, bad1, bad2, obs2_of_1=None):
    """KF1 -> slot 0, KF2 -> slot 1 of one keyframe store; map points 1000 + i (KF1's) and 500000 + i (KF2's) in one map-point store"""
    k1, k2 = sc["kf1"], sc["kf2"]
    KF = corb.KeyFrameStore(3, n + 3); MP = corb.MapPointStore(2 * n, 4)
    for slot, k, T, kid in ((0, k1, sc["T1w"], 11), (1, k2, sc["T2w"], 22)):
        KF.put(slot, k["keys_un"], k["desc"], k["u_right"], None, keyframe_id=kid)
        KF.set_meta(slot, id=kid, client_id=1, flags=0, fx=k["fx"], fy=k["fy"], cx=k["cx"], cy=k["cy"], bf=k["bf"], nlevels=8, Tcw=np.asarray(T, np.float32).reshape(16),
                    inv_level_sigma2=np.concatenate([k["inv_level_sigma2"], np.zeros(8, np.float32)]))
    KF.set_map_points(0, ids1); KF.set_map_points(1, ids2)
    rec = np.zeros(2 * n, corb.MP_RECORD_DTYPE)
    for base, pts, desc, bad, kid, first in ((0, sc["pts1"], sc["desc1"], bad1, 11, 1000), (n, sc["pts2"], sc["desc2"], bad2, 22, 500000)):
        r = rec[base: base + n]
        r["id"] = first + np.arange(n); r["ref_kf_id"] = kid; r["descriptor"] = desc; r["client_id"] = 1
        r["world_pos"] = pts["world"]; r["normal"] = pts["normal"]; r["min_distance"] = pts["min_distance"]; r["max_distance"] = pts["max_distance"]
        r["flags"] = np.where(bad, corb.MP_BAD, 0)
    # observation lists: KF1's point i is seen by KF1 at feature i (+ by KF2 at obs2_of_1[i] when >= 0); KF2's point i by KF2 at feature i
    okf, oidx, off = [], [], [0]
    for i in range(n):
        okf.append(11); oidx.append(i)
        if obs2_of_1 is not None and obs2_of_1[i] >= 0:
            okf.append(22); oidx.append(int(obs2_of_1[i]))
        off.append(len(okf))
    for i in range(n):
        okf.append(22); oidx.append(i); off.append(len(okf))
    rec["n_obs"] = np.diff(off)
    MP.put(0, rec, np.array(off, np.int32), np.array(okf, np.uint64), np.array(oidx, np.uint32))
    MP.build_index(0, 2 * n)
    k = k2
    cam = corb.TrackCamera.make(k["fx"], k["fy"], k["cx"], k["cy"], k["bf"], k["bf"] / k["fx"], k["min_x"], k["max_x"], k["min_y"], k["max_y"], k["scale"])
    return KF, MP, cam


@pytest.mark.parametrize("seed,n,span", [(5310, 2000, 1.0), (5311, 2000, 0.25), (5312, 500, 1.0)])
def test_reloc_projection_on_records(corb, pyorc, synth, seed, n, span):
    """CurrentFrame = KF2's features as a frame record, pKF = KF1's record.  Pointer-level cases built into the records: features of pKF without a MapPoint, with a
    bad one, with one the frame already holds (sAlreadyFound); frame features that hold a MapPoint (skipped as candidates)."""
    rng = np.random.default_rng(seed)
    sc = synth.keyframe_scene(seed, n=n, span=span)
    pts = sc["pts1"]
    has = pts["valid"] != 0                                           # the scene's validity = "pKF's feature holds a usable MapPoint"; split the invalid ones into the three causes
    cause = rng.integers(0, 3, n)                                     # 0: no MapPoint, 1: bad, 2: already found by the frame
    ids1 = np.where(has | (cause != 0), np.uint64(1000) + np.arange(n, dtype=np.uint64), NONE)
    bad1 = ~has & (cause == 1)
    found = np.nonzero(~has & (cause == 2))[0]
    # the frame (KF2's features): claimed2 features hold a MapPoint -- the first len(found) of them hold pKF's "already found" points, the rest unrelated ids
    claimed = sc["claimed2"] != 0
    ids2 = np.where(claimed, np.uint64(700000) + np.arange(n, dtype=np.uint64), NONE)
    ci = np.nonzero(claimed)[0]
    assert len(ci) >= len(found) > 0
    ids2[ci[: len(found)]] = np.uint64(1000) + found.astype(np.uint64)
    KF, MP, cam = _stores(corb, sc, n, ids1, ids2, bad1, np.zeros(n, bool))
    lsf = sc["kf2"]["log_scale_factor"]
    for check_ori in (True, False):
        for th, dist in ((10.0, 100), (3.0, 64)):
            KF.set_map_points(1, ids2)                                # (the call writes the matches into the frame's record: restore)
            r = pyorc.search_by_projection_reloc(sc["kf2"], sc["claimed2"], sc["T2w"], pts, sc["desc1"], th, dist, int(check_ori))
            g = KF.TrackSearchReloc(1, KF, 0, MP, cam, sc["T2w"], lsf, th, dist, check_ori)
            assert np.array_equal(g[0], r[0]) and g[1] == r[1]
            after = KF.get_map_points(1)
            want = ids2.copy(); m = r[0] >= 0; want[m] = ids1[r[0][m]]
            assert np.array_equal(after, want)                        # CurrentFrame.mvpMapPoints[bestIdx2] = pMP
    assert r[1] > 20
    KF.close(); MP.close()


@pytest.mark.parametrize("seed,n", [(5320, 2000), (5321, 1200)])
def test_search_by_sim3_on_records(corb, pyorc, synth, seed, n):
    """vpMatches12 on entry marks some features of KF1 (and, through the matched point's observation of KF2, features of KF2) as already matched"""
    rng = np.random.default_rng(seed)
    sc = synth.keyframe_scene(seed, n=n)
    p1, p2 = sc["pts1"], sc["pts2"]
    v1, v2 = p1["valid"] != 0, p2["valid"] != 0
    # invalid features: half hold no MapPoint, half a bad one
    c1, c2 = rng.random(n) < 0.5, rng.random(n) < 0.5
    ids1 = np.where(v1 | c1, np.uint64(1000) + np.arange(n, dtype=np.uint64), NONE); bad1 = ~v1 & c1
    ids2 = np.where(v2 | c2, np.uint64(500000) + np.arange(n, dtype=np.uint64), NONE); bad2 = ~v2 & c2
    a = (sc["kf1"], sc["kf2"], sc["T1w"], sc["T2w"], p1, sc["desc1"], p2, sc["desc2"], sc["s12"], sc["R12"], sc["t12"], 7.5)
    KF, MP, cam = _stores(corb, sc, n, ids1, ids2, bad1, bad2)
    lsf = sc["kf1"]["log_scale_factor"]
    r = pyorc.search_by_sim3(*a)
    g = KF.SearchBySim3(0, 1, MP, cam, lsf, sc["T1w"], sc["T2w"], sc["s12"], sc["R12"], sc["t12"], 7.5)
    assert np.array_equal(g[0], r[0]) and g[2] == r[1] and r[1] > 100
    m = r[0] >= 0
    assert np.array_equal(g[1][m], ids2[r[0][m]]) and (g[1][~m] == NONE).all()
    KF.close(); MP.close()
    # second round (LoopClosing::ComputeSim3 calls SearchBySim3 with the matches of the first search in vpMatches12): a third of the found pairs enter as already matched --
    # the KF1 feature by the entry itself, the KF2 feature through the matched MapPoint's observation in KF2
    pre = np.nonzero(m)[0][::3]
    matched_ids = np.full(n, NONE, np.uint64); matched_ids[pre] = ids2[r[0][pre]]
    q1 = p1.copy(); q1["valid"][pre] = 0
    q2 = p2.copy(); q2["valid"][r[0][pre]] = 0
    a2 = (sc["kf1"], sc["kf2"], sc["T1w"], sc["T2w"], q1, sc["desc1"], q2, sc["desc2"], sc["s12"], sc["R12"], sc["t12"], 7.5)
    r2 = pyorc.search_by_sim3(*a2)
    KF, MP, cam = _stores(corb, sc, n, ids1, ids2, bad1, bad2)
    g2 = KF.SearchBySim3(0, 1, MP, cam, lsf, sc["T1w"], sc["T2w"], sc["s12"], sc["R12"], sc["t12"], 7.5, matched12_ids=matched_ids)
    assert np.array_equal(g2[0], r2[0]) and g2[2] == r2[1] and (g2[0][pre] == -1).all()
    # arguments
    with pytest.raises(corb.CorbError):
        KF.SearchBySim3(0, 0, MP, cam, lsf, sc["T1w"], sc["T2w"], sc["s12"], sc["R12"], sc["t12"])
    with pytest.raises(corb.CorbError):
        KF.SearchBySim3(0, 2, MP, cam, lsf, sc["T1w"], sc["T2w"], sc["s12"], sc["R12"], sc["t12"])          # an empty slot
    KF.close(); MP.close()


@pytest.mark.parametrize("seed,n,span", [(5330, 2000, 1.0), (5331, 2000, 0.25), (5332, 600, 0.5)])
def test_search_by_projection_scw_on_records(corb, pyorc, synth, seed, n, span):
    """SearchByProjection(KeyFrame* pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:425-538) on records: pKF = KF2's record, vpPoints = KF1's map points by slot (in a
    shuffled order -- the call is order dependent), vpMatched = ids per feature of pKF.  Pointer-level cases in the records: bad points, points already in vpMatched
    (spAlreadyFound), features that hold an id on entry."""
    rng = np.random.default_rng(seed)
    sc = synth.crowd_keyframe_scene(synth.keyframe_scene(seed, n=n, span=span), seed)
    pts = sc["pts1"]
    ok = pts["valid"] != 0
    cause = rng.integers(0, 2, n)                                      # the scene's invalid points: 0 = bad, 1 = already found (held by a feature of pKF on entry)
    claimed = sc["claimed2"] != 0
    ci = np.nonzero(claimed)[0]
    found = np.nonzero(~ok & (cause == 1))[0][: len(ci)]                # (as many as there are features to hold them; the other invalid points are bad)
    bad1 = ~ok; bad1[found] = False
    ids1 = np.uint64(1000) + np.arange(n, dtype=np.uint64)
    assert len(found) > 0
    matched = np.where(claimed, np.uint64(700000) + np.arange(n, dtype=np.uint64), NONE)      # vpMatched on entry: unrelated ids ...
    matched[ci[: len(found)]] = ids1[found]                                                     # ... and the "already found" points of vpPoints
    KF, MP, cam = _stores(corb, sc, n, np.full(n, NONE, np.uint64), np.full(n, NONE, np.uint64), bad1, np.zeros(n, bool))
    lsf = sc["kf2"]["log_scale_factor"]
    order = rng.permutation(n).astype(np.int32)                        # vpPoints[i] = the record in slot order[i]
    total = 0
    for th, s_ in ((10.0, 1.0), (4.0, 1.03)):
        S = sc["T2w"].copy(); S[:3, :] *= np.float32(s_)
        r = pyorc.search_by_projection_scw(sc["kf2"], claimed.astype(np.uint8), S, pts[order], sc["desc1"][order], th)
        ids_after, m, cnt = KF.SearchByProjectionScw(1, MP, order, cam, S, lsf, matched, th)
        assert np.array_equal(m, r[0]) and cnt == r[1]
        want = matched.copy(); hit = r[0] >= 0; want[hit] = ids1[order[r[0][hit]]]
        assert np.array_equal(ids_after, want)                         # vpMatched[bestIdx] = pMP
        assert not np.isin(ids1[found], ids_after[~claimed]).any() and not np.isin(ids1[bad1], ids_after).any()
        total += cnt
    assert total > 40
    # arguments / empty inputs
    ids_after, m, cnt = KF.SearchByProjectionScw(1, MP, order[:0], cam, S, lsf, matched, 10.0)
    assert cnt == 0 and (m == -1).all() and np.array_equal(ids_after, matched)
    with pytest.raises(RuntimeError):
        KF.SearchByProjectionScw(2, MP, order, cam, S, lsf, matched[:0], 10.0)                 # an empty slot
    with pytest.raises(corb.CorbError):
        KF.SearchByProjectionScw(1, MP, np.array([2 * n], np.int32), cam, S, lsf, matched, 10.0)   # a slot outside the map
    KF.close(); MP.close()


def test_mappoint_replace_on_records(corb, pyorc, synth):
    """MapPoint::Replace(pMP) (C/src/MapPoint.cc:277-316) on records against the oracle's restatement (oracle/orc_map.c orc_mappoint_replace + orc_distinctive_descriptors),
    20 random pairs of observation lists: observations that move (the keyframe's match is re-pointed, the entry lands at its place in pMP's ascending list), observations
    pMP already has in the same keyframe (the keyframe's match is erased), keyframes that are not in the store (lists re-linked, no record touched), bad keyframes (they
    keep their matches re-pointed but give no descriptor), a keyframe with mnId 0 next to never-filled slots, full lists (CORB_ERR_CAPACITY, nothing written),
    the same point (no-op); mnFound / mnVisible added, mpReplaced set, this bad and empty."""
    rng = np.random.default_rng(77)
    NKF, F, O = 12, 40, 8
    kf_ids = [0, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377]             # (mnId 0: the first keyframe of a client)
    outside = [610, 987]                                                  # observers that are not in the store
    bad_kf = {21, 144}
    n_full = n_moved = n_erased = 0
    for case in range(20):
        KF = corb.KeyFrameStore(NKF + 4, F); MP = corb.MapPointStore(4, O)          # slots NKF .. NKF+3 are never filled
        descs = {}
        for s_, kid in enumerate(kf_ids):
            kp = np.zeros(F, corb.KP_DTYPE); kp["x"] = rng.uniform(0, 1000, F)
            d = rng.integers(0, 256, (F, 32), dtype=np.uint8); descs[kid] = d
            KF.put(s_, kp, d, None, None, keyframe_id=kid)
            KF.set_meta(s_, id=kid, client_id=1, flags=(corb.KF_BAD if kid in bad_kf else 0), fx=700.0, fy=700.0, cx=600.0, cy=180.0, bf=380.0, nlevels=8, Tcw=np.eye(4, dtype=np.float32).reshape(16))
        pool = kf_ids + outside
        def rand_obs(k):
            ks = sorted(rng.choice(len(pool), k, replace=False).tolist())
            return [(pool[j], int(rng.integers(0, F))) for j in ks]
        na = int(rng.integers(0, O + 1)); nb = int(rng.integers(0, O + 1))
        if case == 5: na, nb = O, O                                       # full lists
        if case == 6: na, nb = O, 0
        obs = {100: rand_obs(na), 200: rand_obs(nb), 300: rand_obs(2)}
        rec = np.zeros(3, corb.MP_RECORD_DTYPE); rec["id"] = [100, 200, 300]; rec["ref_kf_id"] = [3, 5, 3]; rec["client_id"] = 1
        rec["descriptor"] = rng.integers(0, 256, (3, 32), dtype=np.uint8); rec["world_pos"] = rng.normal(0, 3, (3, 3))
        off, okf, oidx = [0], [], []
        for pid in (100, 200, 300):
            okf += [a for a, _ in obs[pid]]; oidx += [b for _, b in obs[pid]]; off.append(len(okf))
        rec["n_obs"] = np.diff(off)
        MP.put(0, rec, np.array(off, np.int32), np.array(okf, np.uint64), np.array(oidx, np.uint32))
        cnt = rng.integers(0, 50, (3, 2))
        MP.set_counters(0, cnt[:, 0].tolist(), cnt[:, 1].tolist())
        held = {}
        for pid in (300, 200, 100):                                       # (where two points name one feature the later write stays: 100's own features hold 100)
            for kid, idx in obs[pid]:
                if kid in kf_ids: held[(kid, idx)] = pid
        for s_, kid in enumerate(kf_ids):
            full = np.full(F, NONE, np.uint64)
            for (k_, idx), pid in held.items():
                if k_ == kid: full[idx] = pid
            KF.set_map_points(s_, full)
        st, into, act, cinto = pyorc.mappoint_replace(100, 200, obs[100], obs[200], O, tuple(cnt[0]), tuple(cnt[1]))
        before = [MP.get(0, 3), MP.get_counters(0, 3), [KF.get_map_points(s_).copy() for s_ in range(NKF)]]
        if st == -1:
            n_full += 1
            with pytest.raises(corb.CorbError, match="no room"):
                MP.Replace(0, 1, KF, 0, NKF + 4)
            r, k, i_ = MP.get(0, 3)
            assert r.tobytes() == before[0][0].tobytes() and np.array_equal(k, before[0][1]) and np.array_equal(i_, before[0][2])      # nothing was written
            assert all(np.array_equal(KF.get_map_points(s_), before[2][s_]) for s_ in range(NKF))
            KF.close(); MP.close(); continue
        assert st == 0 and MP.Replace(0, 1, KF, 0, NKF + 4) == 0
        exp_held = dict(held)
        for (kid, idx), a in zip(obs[100], act):
            if kid in kf_ids:
                if a == 1: exp_held[(kid, idx)] = 200; n_moved += 1
                else: exp_held.pop((kid, idx), None); n_erased += 1
        r, k, i_ = MP.get(0, 3); c = MP.get_counters(0, 3)
        assert r["n_obs"][0] == 0 and (r["flags"][0] & corb.MP_BAD) and c["replaced_by"][0] == 201 and not k[0].any()
        assert r["n_obs"][1] == len(into) and [(int(x), int(y)) for x, y in zip(k[1, : len(into)], i_[1, : len(into)])] == into
        assert (c["n_visible"][1], c["n_found"][1]) == cinto and c["replaced_by"][1] == 0 and not (r["flags"][1] & corb.MP_BAD)
        rows = [descs[kid][idx] for kid, idx in into if kid in kf_ids and kid not in bad_kf]      # pMP->ComputeDistinctiveDescriptors(): non-bad keyframes of the store
        if rows:
            best = pyorc.distinctive_descriptors(np.stack(rows), np.array([0, len(rows)], np.int32))[0]
            assert np.array_equal(r["descriptor"][1], rows[best])
        else:
            assert np.array_equal(r["descriptor"][1], rec["descriptor"][1])
        assert r[2].tobytes() == rec[2].tobytes() and (c["n_visible"][2], c["n_found"][2]) == tuple(cnt[2])       # a bystander
        for s_, kid in enumerate(kf_ids):
            a = KF.get_map_points(s_)
            for idx in range(len(a)):
                assert int(a[idx]) == exp_held.get((kid, idx), int(NONE)), (case, kid, idx)
        assert MP.Replace(1, 1, KF, 0, NKF + 4) == 1                      # the same point
        KF.close(); MP.close()
    assert n_full >= 1 and n_moved > 20 and n_erased > 5
