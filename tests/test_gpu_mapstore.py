"""GPU tests of the complete map transport and of BASELINE configs[3]'s server leg end to end: per-client keyframe + map-point stores ->
corb_map_push_ex over the in-process transport with one host thread per rank (the N-rank bookkeeping on ONE GPU) -> corb_rebase_map_store ->
corb_ba_solve_store, against the oracle on the equivalent flattened arrays (1e-4, BASELINE.json north_star)."""
import threading
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4
CAMS = [(718.856, 718.856, 607.1928, 185.2157, 386.1448), (707.0912, 707.0912, 601.8873, 183.1104, 379.8145)]      # KITTI00-02.yaml / KITTI04-12.yaml


def _rigid(rng):
    T = np.eye(4, dtype=np.float32); Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    T[:3, :3] = Q; T[:3, 3] = rng.normal(0, 3, 3)
    return T


def _fill(corb, cm, F, O, kf_cap, mp_cap):
    """one client's stores from synth.client_maps' description"""
    kf = corb.KeyFrameStore(kf_cap, F); mp = corb.MapPointStore(mp_cap, O)
    for s, k in enumerate(cm["kf"]):
        kf.put(s, k["kp"], k["desc"], k["ur"], None, keyframe_id=k["id"])
        cam = k["cam"]
        kf.set_meta(s, id=k["id"], client_id=k["client_id"], flags=0, fx=cam[0], fy=cam[1], cx=cam[2], cy=cam[3], bf=cam[4], nlevels=8, Tcw=k["Tcw"].reshape(16),
                    inv_level_sigma2=np.concatenate([k["inv_level_sigma2"], np.zeros(8, np.float32)]))
        kf.set_map_points(s, k["mp_id"])
    mp.put(0, cm["mp_records"], cm["obs_off"], cm["obs_kf"], cm["obs_idx"])
    return kf, mp


def _threads(fns):
    """run fns[r]() on one thread per rank; returns results / exceptions in rank order (a hang fails the test instead of blocking it)"""
    out = [None] * len(fns)
    def run(r):
        try:
            out[r] = ("ok", fns[r]())
        except Exception as e:                                       # noqa: BLE001
            out[r] = ("err", e)
    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(len(fns))]
    for t in th: t.start()
    for t in th: t.join(60)
    assert not any(t.is_alive() for t in th), "a rank is stuck in the collective"
    return out


def test_records_round_trip(corb, synth):
    rng = np.random.default_rng(31)
    kf = corb.KeyFrameStore(3, 256)
    n = 200
    kp = np.zeros(n, corb.KP_DTYPE); kp["x"] = rng.uniform(0, 1241, n); kp["octave"] = rng.integers(0, 8, n)
    kf.put(1, kp, rng.integers(0, 256, (n, 32), dtype=np.uint8), None, None, keyframe_id=77)
    m0 = kf.get_meta(1)
    assert m0["id"] == 77 and m0["flags"] == 0 and np.array_equal(m0["Tcw"].reshape(4, 4), np.eye(4, dtype=np.float32))
    assert (kf.get_map_points(1) == corb.NO_MAP_POINT).all()
    T = _rigid(rng); ids = rng.integers(1, 1 << 40, n).astype(np.uint64); ids[::7] = corb.NO_MAP_POINT
    kf.set_meta(1, id=2000077, client_id=3, flags=corb.KF_FIXED, fx=700.5, fy=701.5, cx=600.25, cy=180.75, bf=380.0, nlevels=8, Tcw=T.reshape(16), ba_global_for_kf=9)
    kf.set_map_points(1, ids)
    m = kf.get_meta(1)
    assert m["id"] == 2000077 and kf.get(1)["id"] == 2000077 and m["client_id"] == 3 and m["flags"] == corb.KF_FIXED and m["ba_global_for_kf"] == 9
    assert np.array_equal(m["Tcw"], T.reshape(16)) and m["fx"] == np.float32(700.5) and m["bf"] == np.float32(380.0)
    assert np.array_equal(kf.get_map_points(1), ids) and kf.get(1)["kp"].tobytes() == kp.tobytes()
    mp = corb.MapPointStore(50, 8)
    assert mp.record_bytes() % 64 == 0
    rec = np.zeros(20, corb.MP_RECORD_DTYPE); rec["id"] = np.arange(20) + 1000001; rec["world_pos"] = rng.normal(0, 10, (20, 3)); rec["descriptor"] = rng.integers(0, 256, (20, 32))
    rec["normal"] = rng.normal(0, 1, (20, 3)); rec["flags"][3] = corb.MP_BAD; rec["ref_kf_id"] = 5; rec["min_distance"] = 0.5; rec["max_distance"] = 40
    cnt = rng.integers(0, 9, 20); off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    okf = rng.integers(1, 1 << 40, off[-1]).astype(np.uint64); oi = rng.integers(0, 2000, off[-1]).astype(np.uint32)
    mp.put(10, rec, off, okf, oi)
    g, gk, gi = mp.get(10, 20)
    rec["n_obs"] = cnt
    assert g.tobytes() == rec.tobytes()
    for i in range(20):
        assert np.array_equal(gk[i, : cnt[i]], okf[off[i]: off[i + 1]]) and np.array_equal(gi[i, : cnt[i]], oi[off[i]: off[i + 1]]) and not gk[i, cnt[i]:].any()
    e, _, _ = mp.get(0, 2)
    assert (e["n_obs"] == 0).all()
    with pytest.raises(corb.CorbError):                                  # more observations than a record holds: loud, nothing written
        mp.put(0, rec[:1], np.array([0, 9], np.int32), np.zeros(9, np.uint64), np.zeros(9, np.uint32))
    with pytest.raises(corb.CorbError):
        mp.put(45, rec, off, okf, oi)
    with pytest.raises(corb.CorbError):                                  # malformed FeatureVector offsets are rejected
        kf.set_bow(1, (np.array([3, 10], np.uint32), np.array([0, 5, 3], np.int32), np.arange(5, dtype=np.uint32)))
    kf.close(); mp.close()


def test_map_point_scratch_round_trip(corb):
    """CorbMapPointScratch: set / get are exact, a put of the record clears it (like the counters), neighbours and the record's header / lists are untouched"""
    rng = np.random.default_rng(5)
    MP = corb.MapPointStore(10, 5)
    rec = np.zeros(10, corb.MP_RECORD_DTYPE); rec["id"] = np.arange(10) + 7; rec["world_pos"] = rng.normal(0, 3, (10, 3))
    off = (np.arange(11) * 2).astype(np.int32)
    MP.put(0, rec, off, rng.integers(1, 99, 20).astype(np.uint64), rng.integers(0, 50, 20).astype(np.uint32))
    before = MP.get(0, 10)
    assert not MP.get_scratch(0, 10).tobytes().strip(b"\0")
    sc = np.zeros(4, corb.MP_SCRATCH_DTYPE)
    sc["first_kf_id"] = [-3, 5, 7, 2 ** 40]; sc["track_proj_x"] = [1.5, -2.25, 3e5, 0]; sc["track_in_view"] = [1, 0, 1, 1]; sc["corrected_reference"] = [2 ** 63, 1, 2, 3]; sc["n_obs_weight"] = [2, 4, 6, 7]
    MP.set_scratch(3, sc)
    assert MP.get_scratch(3, 4).tobytes() == sc.tobytes()
    assert not MP.get_scratch(0, 3).tobytes().strip(b"\0") and not MP.get_scratch(7, 3).tobytes().strip(b"\0")
    after = MP.get(0, 10)
    assert all(a.tobytes() == b.tobytes() for a, b in zip(before, after))
    MP.put(4, rec[4:5], np.array([0, 2], np.int32), np.array([1, 2], np.uint64), np.array([3, 4], np.uint32))
    g = MP.get_scratch(3, 4)
    assert g[0].tobytes() == sc[0].tobytes() and not g[1].tobytes().strip(b"\0") and g[2].tobytes() == sc[2].tobytes()
    assert MP.record_bytes() >= 128 + 5 * 12 + 104
    with pytest.raises(corb.CorbError):
        MP.get_scratch(8, 5)
    MP.close()


def test_four_rank_push_on_one_gpu(corb, synth):
    """corb_map_push_ex with world = 4 over the in-process transport: keyframe and map-point records of four client stores arrive on the root, bit for bit,
    at dst_first[r]; nothing else on the root changes; ragged counts incl. an empty rank and a non-contiguous slot list"""
    rng = np.random.default_rng(32)
    W = 4; F = 128; O = 8
    kfs = [corb.KeyFrameStore(24, F) for _ in range(W)]; mps = [corb.MapPointStore(64, O) for _ in range(W)]
    ref = {}
    for r in range(W):
        for s in range(5):
            n = 60 + 10 * s + r
            kp = np.zeros(n, corb.KP_DTYPE); kp["x"] = rng.uniform(0, 1000, n); kp["angle"] = rng.uniform(0, 360, n)
            desc = rng.integers(0, 256, (n, 32), dtype=np.uint8); ur = rng.uniform(-1, 900, n).astype(np.float32)
            kfs[r].put(s, kp, desc, ur, None, keyframe_id=1000000 * r + s + 1)
            kfs[r].set_meta(s, id=1000000 * r + s + 1, client_id=r + 1, Tcw=_rigid(rng).reshape(16), fx=700 + r, nlevels=8)
            ids = rng.integers(1, 1 << 30, n).astype(np.uint64); kfs[r].set_map_points(s, ids)
            fv = synth.feature_vector(n, 6, rng); kfs[r].set_bow(s, fv)
            ref[(r, s)] = (kp, desc, ur, ids, kfs[r].get_meta(s), fv)
        rec = np.zeros(12, corb.MP_RECORD_DTYPE); rec["id"] = 1000000 * r + np.arange(12) + 1; rec["world_pos"] = rng.normal(0, 5, (12, 3)); rec["client_id"] = r + 1
        off = (np.arange(13) * 3).astype(np.int32)
        mps[r].put(0, rec, off, rng.integers(1, 1 << 30, 36).astype(np.uint64), rng.integers(0, 100, 36).astype(np.uint32))
        # the rest of MapPoint's serialised state (MapPoint.h:52-72): counters in the header's spare bytes, tracking / mapping scratch behind the observation lists
        mps[r].set_counters(0, rng.integers(0, 99, 12), rng.integers(0, 99, 12), rng.integers(0, 1 << 20, 12))
        sc = np.zeros(12, corb.MP_SCRATCH_DTYPE)
        for name in sc.dtype.names:
            if name != "pad":
                sc[name] = rng.integers(0, 100, 12) if sc.dtype[name].kind in "iu" else rng.normal(0, 50, 12)
        mps[r].set_scratch(0, sc)
    send_kf = [[0, 1], [4, 2, 0], [], [1, 2, 3]]; send_mp = [[0, 1, 2, 3], list(range(12)), [5], []]
    kf_dst = [10, 12, 15, 15]; mp_dst = [20, 24, 36, 37]
    before_kf = [kfs[0].get(s) for s in range(10)]
    comms = corb.Comm.local(W)
    res = _threads([lambda r=r: comms[r].map_push_ex(kfs[r], send_kf[r], mps[r], send_mp[r], root=0, kf_dst_first=kf_dst, mp_dst_first=mp_dst) for r in range(W)])
    assert all(x[0] == "ok" for x in res), res
    kc, mc = res[0][1]
    assert list(kc) == [2, 3, 0, 3] and list(mc) == [4, 12, 1, 0] and all(res[r][1] is None for r in (1, 2, 3))
    for r in range(W):
        for i, s in enumerate(send_kf[r]):
            g = kfs[0].get(kf_dst[r] + i); kp, desc, ur, ids, meta, fv = ref[(r, s)]
            assert g["id"] == 1000000 * r + s + 1 and g["kp"].tobytes() == kp.tobytes() and np.array_equal(g["desc"], desc) and np.array_equal(g["u_right"], ur)
            assert np.array_equal(kfs[0].get_map_points(kf_dst[r] + i), ids) and kfs[0].get_meta(kf_dst[r] + i).tobytes() == meta.tobytes()
            assert all(np.array_equal(x, np.asarray(y)) for x, y in zip(g["fv"], fv))
        if send_mp[r]:
            g, gk, gi = mps[0].get(mp_dst[r], len(send_mp[r])); o, ok, oi = mps[r].get(0, 12) if r else (None, None, None)
            if r:
                assert g.tobytes() == o[send_mp[r]].tobytes() and np.array_equal(gk, ok[send_mp[r]]) and np.array_equal(gi, oi[send_mp[r]])
                # ... and every other serialised field of the MapPoint travelled with the record
                assert mps[0].get_counters(mp_dst[r], len(send_mp[r])).tobytes() == mps[r].get_counters(0, 12)[send_mp[r]].tobytes()
                assert mps[0].get_scratch(mp_dst[r], len(send_mp[r])).tobytes() == mps[r].get_scratch(0, 12)[send_mp[r]].tobytes()
                assert mps[r].get_scratch(0, 12)["last_frame_seen"].any()
    for s in range(10):                                                   # the root's own slots outside the destination ranges are untouched
        a = kfs[0].get(s)
        assert a["id"] == before_kf[s]["id"] and a["kp"].tobytes() == before_kf[s]["kp"].tobytes()
    # ---- errors are collective: every rank returns the same code, nobody hangs, nothing moves ----
    snap = kfs[0].get(10)["kp"].tobytes()
    res = _threads([lambda r=r: comms[r].map_push_ex(kfs[r], send_kf[r], mps[r], send_mp[r], root=0, kf_dst_first=[10, 12, 15, 22], mp_dst_first=mp_dst) for r in range(W)])
    assert all(x[0] == "err" and "(-2)" in str(x[1]) for x in res), res          # rank 3's three keyframes do not fit at slot 22 of 24: CORB_ERR_CAPACITY everywhere
    res = _threads([lambda r=r: comms[r].map_push_ex(kfs[r], [0, 99] if r == 2 else send_kf[r], mps[r], send_mp[r], root=0, kf_dst_first=kf_dst, mp_dst_first=mp_dst) for r in range(W)])
    assert all(x[0] == "err" and "(-1)" in str(x[1]) for x in res), res          # a non-root rank's bad slot: CORB_ERR_ARG everywhere
    small = corb.KeyFrameStore(4, 64)                                            # a rank whose store has another record size
    small.put(0, np.zeros(3, corb.KP_DTYPE), np.zeros((3, 32), np.uint8))
    res = _threads([lambda r=r: comms[r].map_push_ex(small if r == 1 else kfs[r], [0], None, [], root=0, kf_dst_first=[10, 11, 12, 13]) for r in range(W)])
    assert all(x[0] == "err" and "(-1)" in str(x[1]) for x in res), res
    assert kfs[0].get(10)["kp"].tobytes() == snap
    # ... and the communicator is still usable afterwards; source and destination slots of the root may overlap (records are staged)
    res = _threads([lambda r=r: comms[r].map_push_ex(kfs[r], [1, 0] if r == 0 else [], None, [], root=0, kf_dst_first=[0, 2, 2, 2]) for r in range(W)])
    assert all(x[0] == "ok" for x in res), res
    assert kfs[0].get(0)["id"] == 2 and kfs[0].get(1)["id"] == 1
    for c in comms: c.close()
    small.close()
    for s in kfs + mps: s.close()


def test_asynchronous_push_setup_begin_wait(corb, synth):
    """corb_map_push_setup / _begin / _wait: the root's layout shared once, then pushes with ONE header all-gather whose records arrive as with corb_map_push_ex; the
    verdict of a push that does not fit is the same on every rank without a second round; over the in-process transport with four ranks and over RCCL with one."""
    rng = np.random.default_rng(77)
    W = 4; F = 64; O = 4
    kfs = [corb.KeyFrameStore(16, F) for _ in range(W)]; mps = [corb.MapPointStore(32, O) for _ in range(W)]
    for r in range(W):
        for s_ in range(4):
            n = 30 + s_ + r
            kp = np.zeros(n, corb.KP_DTYPE); kp["x"] = rng.uniform(0, 1000, n)
            kfs[r].put(s_, kp, rng.integers(0, 256, (n, 32), dtype=np.uint8), None, None, keyframe_id=1000 * r + s_ + 1)
        rec = np.zeros(8, corb.MP_RECORD_DTYPE); rec["id"] = 1000 * r + np.arange(8) + 1; rec["world_pos"] = rng.normal(0, 5, (8, 3))
        mps[r].put(0, rec, (np.arange(9) * 2).astype(np.int32), rng.integers(1, 1 << 30, 16).astype(np.uint64), rng.integers(0, 30, 16).astype(np.uint32))
    kf_dst = [8, 10, 12, 14]; mp_dst = [16, 20, 24, 28]
    comms = corb.Comm.local(W)
    res = _threads([lambda r=r: comms[r].map_push_setup(0, kfs[r] if r == 0 else None, mps[r] if r == 0 else None, kf_dst if r == 0 else None, mp_dst if r == 0 else None) for r in range(W)])
    assert all(x[0] == "ok" for x in res), res
    send_kf = [[1], [0, 3], [], [2, 1]]; send_mp = [[0, 1], [7], [2, 3, 4], []]
    def push(r):
        comms[r].map_push_begin(kfs[r], send_kf[r], mps[r], send_mp[r], root=0)
        return comms[r].map_push_wait()
    res = _threads([lambda r=r: push(r) for r in range(W)])
    assert all(x[0] == "ok" for x in res), res
    kc, mc = res[0][1]
    assert list(kc) == [1, 2, 0, 2] and list(mc) == [2, 1, 3, 0]
    for r in range(1, W):
        for i, s_ in enumerate(send_kf[r]):
            assert kfs[0].get(kf_dst[r] + i)["id"] == 1000 * r + s_ + 1 and kfs[0].get(kf_dst[r] + i)["kp"].tobytes() == kfs[r].get(s_)["kp"].tobytes()
        if send_mp[r]:
            g, gk, gi = mps[0].get(mp_dst[r], len(send_mp[r])); o, ok, oi = mps[r].get(0, 8)
            assert g.tobytes() == o[send_mp[r]].tobytes() and np.array_equal(gk, ok[send_mp[r]])
    # a push that does not fit the shared layout: the same verdict on every rank from the one all-gather, nothing in flight afterwards
    res = _threads([lambda r=r: comms[r].map_push_begin(kfs[r], [0, 1, 2] if r == 3 else [], None, [], root=0) for r in range(W)])
    assert all(x[0] == "err" and "(-2)" in str(x[1]) for x in res), res            # rank 3: three keyframes from slot 14 of 16
    res = _threads([lambda r=r: push(r) for r in range(W)])                        # and the communicators still work
    assert all(x[0] == "ok" for x in res), res
    for c in comms: c.close()
    # one rank over RCCL: the enqueued form (no host wait inside begin)
    c1 = corb.Comm(corb.Comm.unique_id(), 0, 1)
    if c1 is not None:
        c1.map_push_setup(0, kfs[0], mps[0], [4], [8])
        c1.map_push_begin(kfs[0], [2, 0], mps[0], [1, 3], root=0)
        kc, mc = c1.map_push_wait()
        assert list(kc) == [2] and list(mc) == [2] and kfs[0].get(4)["id"] == 3 and kfs[0].get(5)["id"] == 1
        c1.close()
    for s_ in kfs + mps: s_.close()


@pytest.mark.parametrize("loop_kf", [0, 3000007])
def test_configs3_server_leg_end_to_end(corb, pyorc, synth, loop_kf):
    """BASELINE configs[3]: 4 clients with the two KITTI camera models, each map in its client's own frame -> push to the server rank -> MapFusion's re-basing ->
    fused global BA (10 iterations, non-robust: GlobalOptimize.cpp:444) from the records, against the oracle on the flattened arrays of the same records"""
    rng = np.random.default_rng(33)
    NC, KPC, PPK = 4, 10, 24
    prob = synth.ba_problem(n_clients=NC, kf_per_client=KPC, pts_per_kf=PPK, seed=1104, max_obs=9, window=4)     # (window < keyframes per client: no repeated (keyframe, point) pair)
    prob["intr"] = np.asarray([CAMS[(k // KPC) % 2] for k in range(NC * KPC)], np.float32)
    prob["point_fixed"][5] = 1
    frames = [np.eye(4, dtype=np.float32)] + [_rigid(rng) for _ in range(NC - 1)]
    cms = synth.client_maps(prob, NC, KPC, frames=frames)
    F = max(len(k["kp"]) for cm in cms for k in cm["kf"]) + 3; O = 16
    stores = [_fill(corb, cm, F, O, NC * KPC if c == 0 else KPC, NC * KPC * PPK if c == 0 else KPC * PPK) for c, cm in enumerate(cms)]
    # a bad keyframe and a bad map point on client 2 (Optimizer.cc:86-87, 108-109, 131-132): no vertex, its observations are no edges
    stores[2][0].set_meta(3, flags=corb.KF_BAD)
    r2, _, _ = stores[2][1].get(7, 1); r2["flags"] |= corb.MP_BAD
    rr, ok_, oi_ = stores[2][1].get(7, 1); n7 = int(rr["n_obs"][0]); stores[2][1].put(7, r2, np.array([0, n7], np.int32), ok_[0, :n7], oi_[0, :n7])
    comms = corb.Comm.local(NC)
    kf_dst = [0, KPC, 2 * KPC, 3 * KPC]; mp_dst = [0, KPC * PPK, 2 * KPC * PPK, 3 * KPC * PPK]
    res = _threads([lambda c=c: comms[c].map_push_ex(stores[c][0], [] if c == 0 else list(range(KPC)), stores[c][1], [] if c == 0 else list(range(KPC * PPK)), root=0,
                                                     kf_dst_first=kf_dst, mp_dst_first=mp_dst) for c in range(NC)])
    assert all(x[0] == "ok" for x in res), res
    KF, MP = stores[0]
    # MapFusion::insertServerMapToGlobleMap per client sub-map, on the records; bit-exact against the oracle's re-basing of the fetched arrays
    for c in range(1, NC):
        ks = list(range(kf_dst[c], kf_dst[c] + KPC)); ms = list(range(mp_dst[c], mp_dst[c] + KPC * PPK))
        P0 = np.stack([KF.get_meta(s)["Tcw"].reshape(4, 4) for s in ks]); X0 = MP.get(ms[0], len(ms))[0]["world_pos"].copy()
        corb.RebaseMapStore(frames[c], KF, ks, MP, ms)
        P1 = np.stack([KF.get_meta(s)["Tcw"].reshape(4, 4) for s in ks]); X1 = MP.get(ms[0], len(ms))[0]["world_pos"]
        rP, rX = pyorc.rebase_map(frames[c], P0, X0)
        assert np.array_equal(P1, rP) and np.array_equal(X1, rX)
        assert np.abs(P1 - prob["poses"][ks]).max() < 1e-4 * max(1, np.abs(prob["poses"]).max())      # ... and it is the fused map again
    # the flattened problem the oracle solves: exactly what the records hold
    K, M = NC * KPC, NC * KPC * PPK
    metas = [KF.get_meta(s) for s in range(K)]
    poses = np.stack([m["Tcw"].reshape(4, 4) for m in metas]); intr = np.array([[m["fx"], m["fy"], m["cx"], m["cy"], m["bf"]] for m in metas], np.float32)
    kbad = np.array([m["flags"] & corb.KF_BAD for m in metas], bool)
    pose_fixed = np.array([(m["id"] == 1) or bool(m["flags"] & corb.KF_FIXED) for m in metas], np.uint8) | kbad.astype(np.uint8)
    rec, okf, oidx = MP.get(0, M)
    id2k = {int(m["id"]): i for i, m in enumerate(metas)}
    kfd = [KF.get(s) for s in range(K)]
    edges = []
    for j in range(M):
        if rec["flags"][j] & corb.MP_BAD:
            continue
        for t in range(rec["n_obs"][j]):
            k = id2k.get(int(okf[j, t]))
            if k is None or kbad[k]:
                continue
            f = int(oidx[j, t]); kp = kfd[k]["kp"][f]
            edges.append((k, j, kp["x"], kp["y"], kfd[k]["u_right"][f], metas[k]["inv_level_sigma2"][kp["octave"]]))
    edges = np.array(edges, corb.EDGE_DTYPE)
    point_fixed = ((rec["flags"] & corb.MP_FIXED) != 0).astype(np.uint8)
    assert len(edges) < len(prob["edges"]) and point_fixed[5 % M] in (0, 1)
    o = pyorc.ba_solve(poses, pose_fixed, rec["world_pos"].copy(), point_fixed, edges, 0, 0, 0, 0, 0, iters=10, robust=False, intr=intr)
    g = corb.GlobalBundleAdjustemntStore(KF, list(range(K)), MP, list(range(M)), nIterations=10, bRobust=False, nLoopKF=loop_kf)
    n_active = int((~((pose_fixed[edges["pose"]] != 0) & (point_fixed[edges["point"]] != 0))).sum())      # allVerticesFixed edges are dropped (sparse_optimizer.cpp:234)
    assert g["iters_done"] == o["iters_done"] and g["trials"] == o["trials"] and g["structure"]["active_edges"] == n_active
    assert np.allclose(g["chi2"], o["chi2"], rtol=RTOL) and g["chi2"][-1] < 0.3 * g["chi2"][0]
    st = max(1.0, np.abs(o["poses"][:, :3, 3]).max()); sp = max(1.0, np.abs(o["points"]).max())
    assert np.abs(g["poses"] - o["poses"]).max() <= RTOL * st and np.abs(g["points"] - o["points"]).max() <= RTOL * sp
    # write-back into the records (Optimizer.cc:216-262): nLoopKF == 0 -> SetPose / SetWorldPos, else mTcwGBA / mPosGBA + mnBAGlobalForKF
    rec1, _, _ = MP.get(0, M)
    for k in range(K):
        m = KF.get_meta(k)
        if kbad[k]:
            assert m.tobytes() == metas[k].tobytes()                       # a bad keyframe is not touched
        elif loop_kf == 0:
            assert np.array_equal(m["Tcw"].reshape(4, 4), g["poses"][k]) and m["ba_global_for_kf"] == 0
        else:
            assert np.array_equal(m["TcwGBA"].reshape(4, 4), g["poses"][k]) and np.array_equal(m["Tcw"], metas[k]["Tcw"]) and m["ba_global_for_kf"] == loop_kf
    touched = np.zeros(M, bool); touched[edges["point"]] = True
    upd = touched & (point_fixed == 0) & ((rec["flags"] & corb.MP_BAD) == 0)
    if loop_kf == 0:
        assert np.array_equal(rec1["world_pos"][upd], g["points"][upd]) and np.array_equal(rec1["world_pos"][~upd], rec["world_pos"][~upd])
    else:
        assert np.array_equal(rec1["pos_gba"][upd], g["points"][upd]) and np.array_equal(rec1["world_pos"], rec["world_pos"]) and (rec1["ba_global_for_kf"][upd] == loop_kf).all()
        assert (rec1["ba_global_for_kf"][~upd] == 0).all()
    assert np.abs(g["poses"][1:] - poses[1:]).max() > 1e-4                  # the estimates moved
    for c in comms: c.close()
    for a, b in stores: a.close(); b.close()


def test_put_batch_equals_single_puts_and_store_gba_equals_array_gba(corb, synth):
    """corb_kf_store_put_batch writes the records corb_kf_store_put_host + set_meta + set_map_points write; and the global BA from store records (graph derived and
    flattened on the device) returns, bit for bit, what corb_ba_solve_ex returns for the flat arrays the records were made from (same lists, same kernels)."""
    KPC, PPK = 30, 40
    prob = synth.ba_problem_fast(n_clients=4, kf_per_client=KPC, pts_per_kf=PPK, seed=1051, obs_range=(3, 8), cams=CAMS)
    ma = synth.map_arrays(prob, KPC, PPK)
    K, M = len(prob["poses"]), len(prob["points"])
    A = corb.KeyFrameStore(K, ma["max_features"]); B = corb.KeyFrameStore(K, ma["max_features"])
    A.put_batch(0, ma["meta"], ma["feat_off"], ma["kp"], None, ma["ur"], None, ma["mp_id"])
    for k in (0, 7, K - 1):
        f0, f1 = ma["feat_off"][k], ma["feat_off"][k + 1]
        B.put(k, ma["kp"][f0:f1], np.zeros((f1 - f0, 32), np.uint8), ma["ur"][f0:f1], None, keyframe_id=int(ma["meta"]["id"][k]))
        m = ma["meta"][k]
        B.set_meta(k, **{n: m[n] for n in m.dtype.names}); B.set_map_points(k, ma["mp_id"][f0:f1])
        a, b = A.get(k), B.get(k)
        assert a["id"] == b["id"] and a["kp"].tobytes() == b["kp"].tobytes() and np.array_equal(a["u_right"], b["u_right"]) and np.array_equal(a["depth"], b["depth"])
        assert A.get_meta(k).tobytes() == B.get_meta(k).tobytes() and np.array_equal(A.get_map_points(k), B.get_map_points(k)) and not a["flags"].any()
    with pytest.raises(corb.CorbError):
        A.put_batch(K - 1, ma["meta"][:2], ma["feat_off"][:3], ma["kp"])
    MP = corb.MapPointStore(M, ma["max_obs"]); MP.put(0, ma["mp_records"], ma["obs_off"], ma["obs_kf"], ma["obs_idx"])
    for solver in (1, 2):
        g = corb.GlobalBundleAdjustemntStore(A, np.arange(K), MP, np.arange(M), nIterations=6, bRobust=False, nLoopKF=99, solver=solver)
        # (a record lists a point's observations in mObservations order -- ascending keyframe id -- so the flat edge array is given in that order too)
        e = prob["edges"]; e = e[np.lexsort((e["pose"], e["point"]))]
        h = corb.Optimizer.GlobalBundleAdjustemnt(prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], e, 0, 0, 0, 0, 0, nIterations=6, bRobust=False,
                                                  solver=solver, intr=prob["intr"])
        assert g["structure"] == h["structure"] and g["iters_done"] == h["iters_done"] and g["trials"] == h["trials"]
        assert np.array_equal(g["chi2"], h["chi2"]) and np.array_equal(g["poses"], h["poses"]) and np.array_equal(g["points"], h["points"])
    A.close(); B.close(); MP.close()
