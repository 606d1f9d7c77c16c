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


# ---- the signature-preserving adapters (corb_adapter_orbslam.hpp) run from C++ on test doubles, compared with the oracle ----
def _rec(f, a):
    b = np.ascontiguousarray(a).tobytes()
    f.write(np.uint32(len(b)).tobytes()); f.write(b)


def _read_records(path):
    raw = open(path, "rb").read(); out = []; o = 0
    while o < len(raw):
        n = int(np.frombuffer(raw, np.uint32, 1, o)[0]); out.append(raw[o + 4:o + 4 + n]); o += 4 + n
    return out


def _write_kf(f, desc, kp, ur, flag, fv):
    _rec(f, desc); _rec(f, kp); _rec(f, ur.astype(np.float32)); _rec(f, flag.astype(np.uint8))
    _rec(f, fv[0].astype(np.uint32)); _rec(f, fv[1].astype(np.int32)); _rec(f, fv[2].astype(np.uint32))


def test_cpp_adapters_match_oracle(tmp_path, corb, pyorc, synth):
    """ORBmatcher::SearchByBoW x3 / SearchForTriangulation and Optimizer::GlobalBundleAdjustemnt / PoseOptimization with the REFERENCE'S SIGNATURES
    (KeyFrame*, Frame&, vector<MapPoint*>&, Cache*, bool* pbStopFlag, nLoopKF): flattening, index -> MapPoint* map-back and the nLoopKF write-back
    are the adapter templates of corb-slam_amd/host/corb_adapter_orbslam.hpp, instantiated on the test doubles of tests/host/mock_orbslam.hpp."""
    from test_oracle_match import _make
    exe = tmp_path / "adapter_main"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "corb-slam_amd", "host"),
                           "-I", os.path.join(ROOT, "tests", "host"), os.path.join(ROOT, "tests", "host", "adapter_main.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "corb-slam_amd"), "-lcorb_accel", "-Wl,-rpath," + os.path.join(ROOT, "corb-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(77)
    scene = tmp_path / "scene.bin"
    with open(scene, "wb") as f:
        # A. BoW: flag 1 = good MapPoint, 2 = bad MapPoint (isBad()), 0 = none
        d1, a1, v1, fv1, d2, a2, v2, fv2 = _make(rng, synth, 700, 650, 30)
        bad1 = (rng.random(700) < 0.1) & (v1 == 0); bad2 = (rng.random(650) < 0.1) & (v2 == 0)
        kp1 = np.zeros(700, corb.KP_DTYPE); kp1["angle"] = a1; kp2 = np.zeros(650, corb.KP_DTYPE); kp2["angle"] = a2
        _write_kf(f, d1, kp1, -np.ones(700), v1 + 2 * bad1, fv1); _write_kf(f, d2, kp2, -np.ones(650), v2 + 2 * bad2, fv2)
        _rec(f, np.float32(0.9)); _rec(f, np.int32(1))
        # B. triangulation
        n1, n2 = 600, 560
        td1 = synth.correlated_descriptors(n1, rng); td2, src = synth.correlated_descriptors(n2, rng, base=td1, flip=0.04)
        tk1 = np.zeros(n1, corb.KP_DTYPE); tk2 = np.zeros(n2, corb.KP_DTYPE)
        tk1["x"], tk1["y"] = rng.uniform(0, 1241, n1), rng.uniform(0, 376, n1)
        tk2["x"] = tk1["x"][src] - rng.uniform(0, 40, n2); tk2["y"] = tk1["y"][src] + rng.normal(0, 0.8, n2)
        tk1["angle"] = rng.uniform(0, 360, n1); tk2["angle"] = (tk1["angle"][src] + rng.normal(0, 15, n2)) % 360
        tk1["octave"] = rng.integers(0, 8, n1); tk2["octave"] = rng.integers(0, 8, n2)
        ur1 = np.where(rng.random(n1) < 0.6, tk1["x"] - 5, -1).astype(np.float32); ur2 = np.where(rng.random(n2) < 0.6, tk2["x"] - 5, -1).astype(np.float32)
        mp1 = (rng.random(n1) < 0.3).astype(np.uint8); mp2 = (rng.random(n2) < 0.3).astype(np.uint8)
        tf1 = synth.feature_vector(n1, 20, rng); tf2 = synth.feature_vector(n2, 20, rng)
        F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-4, (3, 3)).astype(np.float32))
        T2 = np.eye(4, dtype=np.float32); T2[:3, :3] = synth._rot(0.01, -0.02, 0.005).astype(np.float32); T2[:3, 3] = [0.5, 0.02, -0.1]
        Ow1 = np.array([0.1, -0.05, 0.3], np.float32); cam = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)
        scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32); sigma2 = scale * scale
        _write_kf(f, td1, tk1, ur1, mp1, tf1); _write_kf(f, td2, tk2, ur2, mp2, tf2)
        for a in (T2, Ow1, F12, cam, scale, sigma2):
            _rec(f, a)
        _rec(f, np.int32(0))
        # C. global BA: two camera models, one bad keyframe, one bad / one fixed map point, keyframe mnId 1 = index 0
        cams = [synth.KITTI_CAMS["00-02"], synth.KITTI_CAMS["04-12"]]
        p = synth.ba_problem_fast(n_clients=2, kf_per_client=10, pts_per_kf=15, seed=5151, cams=cams, window=3)
        K, M = len(p["poses"]), len(p["points"])
        kf_fixed = np.zeros(K, np.uint8); kf_fixed[4] = 1                      # getFixed() (received from the server)
        kf_bad = np.zeros(K, np.uint8); kf_bad[7] = 1
        mp_fixed = np.zeros(M, np.uint8); mp_fixed[5] = 1
        mp_bad = np.zeros(M, np.uint8); mp_bad[9] = 1
        e = p["edges"]; octv = np.round(np.log(1.0 / e["inv_sigma2"].astype(np.float64)) / np.log(1.44)).astype(np.int32)
        for a in (p["poses"], p["intr"], p["points"], kf_fixed, kf_bad, mp_fixed, mp_bad, e, octv):
            _rec(f, a)
        # D. pose optimisation: 20 % of the features hold no MapPoint
        q = synth.pose_opt_problem(seed=3033, n=260)
        has = (rng.random(260) < 0.8).astype(np.uint8)
        for a in (q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], np.array([q["fx"], q["fy"], q["cx"], q["cy"], q["bf"]], np.float32), has):
            _rec(f, a)
        # E. the map of C once more, for the store adapter (objects -> device records -> global BA on the records -> objects)
        for a in (p["poses"], p["intr"], p["points"], kf_fixed, kf_bad, mp_fixed, mp_bad, e, octv):
            _rec(f, a)
        # F. two frames and a map for the tracking calls on records (FrameStoreT)
        sys_path_tools = os.path.join(ROOT, "tools")
        import sys
        if sys_path_tools not in sys.path:
            sys.path.insert(0, sys_path_tools)
        import replay_client as rc
        wF = rc.World(91, 12); CAM = rc.CAM; f32 = lambda x: np.float32(x)
        nLm = len(wF.X)
        ids = (np.arange(nLm, dtype=np.int64) * 5 + 700)
        C0 = np.array([0.0, 0.0, 4.0], np.float32); PO = wF.Xest - C0; dist = np.linalg.norm(PO, axis=1).astype(np.float32)
        nrm = (PO / dist[:, None]).astype(np.float32); dmax = (dist * wF.scale[wF.octave] * np.float32(1.3)).astype(np.float32); dmin = (dmax / wF.scale[7] / np.float32(1.5)).astype(np.float32)
        mbadF = np.zeros(nLm, np.uint8); mbadF[::19] = 1; mobsF = np.ones(nLm, np.int32); mobsF[::23] = 0
        logs = np.float32(np.log(np.float32(1.2)))
        camv = np.array([f32(CAM["fx"]), f32(CAM["fy"]), f32(CAM["cx"]), f32(CAM["cy"]), f32(CAM["bf"]), f32(CAM["bf"]) / f32(CAM["fx"]), 0.0, CAM["w"], 0.0, CAM["h"], logs], np.float32)
        for a in (camv, wF.scale, ids, wF.Xest, nrm, dmin, dmax, wF.desc, mbadF, mobsF):
            _rec(f, a)
        frF = []
        for t, Tf in ((5, wF.pose(5).astype(np.float32)), (6, None)):
            keys, ur, desc, lm = wF.observe(t)
            if Tf is None:
                Tf = wF.pose(6).astype(np.float32); Tf[0, 3] += np.float32(0.02)          # the predicted pose
                held = np.full(len(lm), -1, np.int32); outl = np.zeros(len(lm), np.uint8)
            else:
                held = np.where(np.arange(len(lm)) % 5 == 0, -1, lm).astype(np.int32); outl = (np.arange(len(lm)) % 29 == 0).astype(np.uint8)
            frF.append(dict(keys=keys, ur=ur, desc=desc, lm=lm, held=held, outl=outl, T=Tf))
            for a in (desc, keys, ur, held, outl, Tf):
                _rec(f, a)
        rngF = np.random.default_rng(6)
        localF = np.unique(np.concatenate([frF[0]["lm"], frF[1]["lm"], rngF.integers(0, nLm, 800)])).astype(np.int32); rngF.shuffle(localF)
        _rec(f, localF)
        # G. the map of C / E once more for LocalBundleAdjustment on records: the first NLOC keyframes local, the others fixed cameras
        NLOC = 6
        for a in (p["poses"], p["intr"], p["points"], kf_fixed, kf_bad, mp_fixed, mp_bad, e, octv, np.array([NLOC], np.int32)):
            _rec(f, a)
        # H. SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th): KF2 of a keyframe scene, KF1's points; the scene's invalid points are bad or already in vpMatched
        scH = synth.keyframe_scene(5177, n=1500, span=0.4); nH = 1500; rngH = np.random.default_rng(5177)
        okH = scH["pts1"]["valid"] != 0; causeH = rngH.integers(0, 2, nH)
        badH = ~okH & (causeH == 0); foundH = np.nonzero(~okH & (causeH == 1))[0]
        clH = np.nonzero(scH["claimed2"])[0]; assert len(clH) >= len(foundH) > 0
        heldH = np.where(scH["claimed2"] != 0, -2, -1).astype(np.int32); heldH[clH[: len(foundH)]] = foundH
        SH = scH["T2w"].copy(); SH[:3, :] *= np.float32(1.02)
        k2 = scH["kf2"]
        camH = np.array([k2["fx"], k2["fy"], k2["cx"], k2["cy"], k2["bf"], k2["min_x"], k2["min_y"], k2["max_x"], k2["max_y"], k2["log_scale_factor"]], np.float32)
        for a in (k2["desc"], k2["keys_un"], k2["u_right"], camH, k2["scale"], k2["inv_level_sigma2"], SH, scH["pts1"]["world"], scH["pts1"]["normal"],
                  scH["pts1"]["min_distance"], scH["pts1"]["max_distance"], scH["desc1"], badH.astype(np.uint8), heldH, np.int32(10)):
            _rec(f, np.ascontiguousarray(a))
        # I. SearchForInitialization(Frame&, Frame&, vbPrevMatched, vnMatches12, windowSize): a monocular initialiser's pair with features that lose their match to later ones
        fI1, fI2, pmI, _ = synth.monocular_init_pair(7300, n=1200, span=0.6, crowd=True, steal_frac=0.15)
        for a in (fI1["desc"], fI1["keys_un"], fI2["desc"], fI2["keys_un"], np.array([fI2["min_x"], fI2["min_y"], fI2["max_x"], fI2["max_y"]], np.float32), pmI, np.int32(100)):
            _rec(f, np.ascontiguousarray(a))
    outp = tmp_path / "out.bin"
    subprocess.check_call([str(exe), str(scene), str(outp)])
    rec = _read_records(outp); I32 = lambda b: np.frombuffer(b, np.int32); F32 = lambda b: np.frombuffer(b, np.float32)
    # A
    r0, n0 = pyorc.search_by_bow(0, d1, a1, v1, pyorc.FeatVec(*fv1), d2, a2, np.ones_like(v2), pyorc.FeatVec(*fv2), 0.9, 1)
    r1, n1_ = pyorc.search_by_bow(1, d1, a1, v1, pyorc.FeatVec(*fv1), d2, a2, v2, pyorc.FeatVec(*fv2), 0.9, 1)
    assert np.array_equal(I32(rec[0]), r0) and np.array_equal(I32(rec[1]), r1) and np.array_equal(I32(rec[2]), r0)
    assert list(I32(rec[3])) == [n0, n1_, n0] and n0 > 0 and n1_ > 0 and (r0 >= 0).sum() == n0
    # B: the epipole as ORBmatcher.cc:799-808 computes it (float matrices, the product accumulated in double)
    C2 = (T2[:3, :3].astype(np.float64) @ Ow1.astype(np.float64)).astype(np.float32) + T2[:3, 3]
    invz = np.float32(1.0) / C2[2]
    ex = cam[0] * C2[0] * invz + cam[2]; ey = cam[1] * C2[1] * invz + cam[3]
    rp, rn = pyorc.search_for_triangulation(td1, tk1, ur1, mp1, pyorc.FeatVec(*tf1), td2, tk2, ur2, mp2, pyorc.FeatVec(*tf2), F12, float(ex), float(ey), scale, sigma2, False, True)
    assert int(I32(rec[5])[0]) == rn and np.array_equal(I32(rec[4]).reshape(-1, 2), np.asarray(rp).reshape(-1, 2)) and rn > 0
    # C: oracle on the flattened problem (bad keyframe / bad point dropped, fixed flags: mnId == 1 or getFixed())
    keep_kf = kf_bad == 0; keep_mp = mp_bad == 0
    kmap = np.cumsum(keep_kf) - 1; mmap = np.cumsum(keep_mp) - 1
    es = e[keep_kf[e["pose"]] & keep_mp[e["point"]]].copy(); es["pose"] = kmap[es["pose"]]; es["point"] = mmap[es["point"]]
    pf = kf_fixed.copy(); pf[0] = 1
    ro = pyorc.ba_solve(p["poses"][keep_kf], pf[keep_kf], p["points"][keep_mp], mp_fixed[keep_mp], es, p["fx"], p["fy"], p["cx"], p["cy"], p["bf"],
                        iters=10, robust=False, intr=p["intr"][keep_kf])
    has_edge = np.zeros(M, bool); has_edge[e["point"][keep_kf[e["pose"]]]] = True
    for pass_, base in ((0, 6), (1, 10)):
        T = F32(rec[base]).reshape(K, 4, 4); X = F32(rec[base + 1]).reshape(M, 3); marks = I32(rec[base + 2]); cnt = I32(rec[base + 3])
        written_kf = keep_kf & (kf_fixed == 0)                                   # mnId 1 is written back too (fixed by id only)
        written_mp = keep_mp & (mp_fixed == 0) & has_edge
        Tref = np.zeros((K, 4, 4), np.float32); Tref[keep_kf] = ro["poses"]
        Xref = np.zeros((M, 3), np.float32); Xref[keep_mp] = ro["points"]
        assert np.abs(T[written_kf] - Tref[written_kf]).max() < 1e-4 and np.abs(X[written_mp] - Xref[written_mp]).max() < 1e-3
        assert np.abs(T[written_kf][1:] - p["poses"][written_kf][1:]).max() > 1e-3            # ... and they did move
        if pass_ == 0:                                                           # nLoopKF == 0: SetPose / SetWorldPos + cache marks + UpdateNormalAndDepth
            assert np.array_equal(T[~written_kf], p["poses"].reshape(K, 4, 4)[~written_kf]) and np.array_equal(X[~written_mp], p["points"][~written_mp])
            assert list(cnt) == [int(written_kf.sum()), int(written_mp.sum())]
            assert np.array_equal(marks[K:] // 1000, written_mp.astype(np.int32)) and not marks[:K].any()
        else:                                                                    # nLoopKF == 7: mTcwGBA / mPosGBA + mnBAGlobalForKF, the map itself untouched
            assert np.all(T[~written_kf] == -777) and np.all(X[~written_mp] == -777) and list(cnt) == [0, 0]
            assert np.array_equal(marks[:K], 7 * written_kf.astype(np.int32)) and np.array_equal(marks[K:], 7 * written_mp.astype(np.int32))
    # D
    sel = has.astype(bool); n = int(sel.sum())
    ed = np.zeros(n, pyorc.EDGE_DTYPE); ed["pose"] = 0; ed["point"] = np.arange(n); ed["u"] = q["obs"][sel, 0]; ed["v"] = q["obs"][sel, 1]; ed["ur"] = q["obs"][sel, 2]; ed["inv_sigma2"] = q["inv_sigma2"][sel]
    rq = pyorc.ba_solve_staged(q["Tcw0"].reshape(1, 16), np.zeros(1, np.uint8), q["points"][sel], np.ones(n, np.uint8), ed, q["fx"], q["fy"], q["cx"], q["cy"], q["bf"], pyorc.POSE_OPT_STAGES)
    Tq = F32(rec[14]).reshape(4, 4); oq = np.frombuffer(rec[15], np.uint8); nq = int(I32(rec[16])[0])
    assert np.abs(Tq - rq["poses"][0]).max() < 1e-4 and np.array_equal(oq[sel], rq["outlier"]) and nq == n - int(rq["outlier"].sum())
    assert np.all(oq[~sel] == 1)                                                 # features without a MapPoint keep their flag (the mock starts them at true)
    # E: the store adapter reaches the same estimates and applies the same write-back policy as the array adapter of C
    for pass_, base in ((0, 17), (1, 21)):
        T = F32(rec[base]).reshape(K, 4, 4); X = F32(rec[base + 1]).reshape(M, 3); marks = I32(rec[base + 2]); cnt = I32(rec[base + 3])
        Ta = F32(rec[6 if pass_ == 0 else 10]).reshape(K, 4, 4); Xa = F32(rec[7 if pass_ == 0 else 11]).reshape(M, 3); ma = I32(rec[8 if pass_ == 0 else 12]); ca = I32(rec[9 if pass_ == 0 else 13])
        written_kf = keep_kf & (kf_fixed == 0); written_mp = keep_mp & (mp_fixed == 0) & has_edge
        assert cnt[2] == 10 and cnt[3] == len(es) - int(((pf[keep_kf][es["pose"]] != 0) & (mp_fixed[keep_mp][es["point"]] != 0)).sum())
        assert np.abs(T[written_kf] - Ta[written_kf]).max() < 2e-5 and np.abs(X[written_mp] - Xa[written_mp]).max() < 2e-4      # (observations in mnId order instead of pointer order: rounding)
        assert np.array_equal(T[~written_kf], Ta[~written_kf]) and np.array_equal(X[~written_mp], Xa[~written_mp])
        assert np.array_equal(marks, ma) and list(cnt[:2]) == list(ca)
    # F: the C++ FrameStoreT calls leave the Frame where the Python-level calls on the same records leave it (tests/test_gpu_track.py holds those against the host-pointer calls)
    NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
    camF = corb.TrackCamera.make(*[float(v) for v in camv[:10]], wF.scale)
    mpF = corb.MapPointStore(nLm, 2); recF = np.zeros(nLm, corb.MP_RECORD_DTYPE)
    recF["id"] = ids.astype(np.uint64); recF["world_pos"] = wF.Xest; recF["normal"] = nrm; recF["min_distance"] = dmin; recF["max_distance"] = dmax; recF["descriptor"] = wF.desc
    recF["flags"] = mbadF; recF["n_obs"] = mobsF
    offF = np.concatenate([[0], np.cumsum(mobsF)]).astype(np.int32)
    mpF.put(0, recF, offF, np.ones(offF[-1], np.uint64), np.zeros(offF[-1], np.uint32)); mpF.build_index(0, nLm)
    kfF = corb.KeyFrameStore(2, 2048)
    inv_s2 = np.zeros(16, np.float32); inv_s2[:8] = (np.float32(1.0) / (wF.scale * wF.scale)).astype(np.float32)
    for slot, fr in enumerate(frF):
        meta = np.zeros((), corb.KF_META_DTYPE); meta["id"] = slot + 1; meta["nlevels"] = 8; meta["inv_level_sigma2"] = inv_s2; meta["Tcw"] = fr["T"].reshape(16)
        for k_, v_ in zip(("fx", "fy", "cx", "cy", "bf"), camv[:5]):
            meta[k_] = v_
        kfF.put_frame(slot, fr["keys"], fr["desc"], fr["ur"], None, meta)
        kfF.set_map_points(slot, np.where(fr["held"] >= 0, ids[np.maximum(fr["held"], 0)].astype(np.uint64), NONE)); kfF.set_flags(slot, (fr["outl"] * 2).astype(np.uint8))
    _, n1 = kfF.TrackSearchLastFrame(1, 0, mpF, frF[1]["T"], frF[0]["T"], camF, 7.0)
    T1, o1, i1 = kfF.TrackPoseOptimization(1, mpF, camF, frF[1]["T"], discard_outliers=True)
    _, n2, nv = kfF.TrackSearchLocalPoints(1, mpF, ids[localF].astype(np.uint64), camF, T1, float(logs), 1.0, 0.8)
    T2f, o2, i2 = kfF.TrackPoseOptimization(1, mpF, camF, T1)
    cntF = I32(rec[25]); heldF = np.frombuffer(rec[26], np.int64); oF = np.frombuffer(rec[27], np.uint8); TF = F32(rec[28])
    assert list(cntF) == [n1, i1, n2, i2, nv] and n1 > 700 and n2 > 20 and nv > 200
    idsF = kfF.get_map_points(1); flF = kfF.get(1)["flags"]
    want_held = np.where((idsF != NONE) & ((flF & 4) == 0), idsF.astype(np.int64), -1)
    assert np.array_equal(heldF, want_held) and np.array_equal(oF.astype(bool), o2) and np.array_equal(TF, np.asarray(T2f).reshape(16)) and o1.sum() > 0
    kfF.close(); mpF.close()
    # G: MapStoreT::LocalBundleAdjustment / ReadBackLocalBA against the oracle's LocalBundleAdjustment on map objects built from the same arrays
    Tg = F32(rec[29]).reshape(K, 4, 4); Xg = F32(rec[30]).reshape(M, 3); erg = I32(rec[31]).reshape(-1, 2); nobs_obj = I32(rec[32]); nobs_rec = I32(rec[33]); heldg = I32(rec[34]); cntg = I32(rec[35])
    feats = [[] for _ in range(K)]
    tab = [np.array([1.0 / 1.44 ** l for l in range(8)], np.float32) for _ in range(K)]
    for ei in range(len(e)):
        k = int(e["pose"][ei]); feats[k].append(ei); tab[k][octv[ei]] = e["inv_sigma2"][ei]
    okfs = []
    for k in range(K):
        keys = np.zeros(len(feats[k]), pyorc.KP_DTYPE); keys["x"] = e["u"][feats[k]]; keys["y"] = e["v"][feats[k]]; keys["octave"] = octv[feats[k]]
        okfs.append(dict(id=k + 1, T=p["poses"][k].reshape(4, 4).copy(), fixed=bool(kf_fixed[k]), bad=bool(kf_bad[k]), keys=keys, ur=e["ur"][feats[k]].astype(np.float32),
                         mp=[1000 + int(e["point"][ei]) for ei in feats[k]], intr=[float(c) for c in p["intr"][k]], nlevels=8, inv_level_sigma2=tab[k]))
    omps = [dict(id=1000 + m, pos=p["points"][m].copy(), fixed=bool(mp_fixed[m]), bad=bool(mp_bad[m]), obs={}, ref=0, nObs=0, normal=np.zeros(3, np.float32), min_distance=np.float32(0), max_distance=np.float32(0)) for m in range(M)]
    for k in range(K):
        for fi, ei in enumerate(feats[k]):
            m = omps[int(e["point"][ei])]; m["obs"][k + 1] = fi; m["nObs"] += 2 if e["ur"][ei] >= 0 else 1
    for m in omps:
        m["ref"] = min(m["obs"]) if m["obs"] else 0
    og = pyorc.local_bundle_adjustment(okfs[:NLOC], okfs[NLOC:], omps, 1.2)
    assert sorted(map(tuple, erg.tolist())) == sorted(og["erase"]) and len(og["erase"]) > 0
    for k in range(K):
        tol = 1e-4 * max(1.0, np.abs(okfs[k]["T"]).max())
        assert np.abs(Tg[k] - okfs[k]["T"]).max() <= tol, k
        if k >= NLOC or kf_fixed[k] or kf_bad[k]:
            assert np.array_equal(Tg[k], p["poses"][k].reshape(4, 4))                  # fixed cameras / getFixed() / bad: not written
        assert heldg[k] == sum(q is not None for q in okfs[k]["mp"]), k
    for m in range(M):
        assert np.abs(Xg[m] - omps[m]["pos"]).max() <= 1e-4 * max(1.0, np.abs(omps[m]["pos"]).max()), m
        want = -1 if omps[m]["bad"] else len(omps[m]["obs"])
        assert nobs_obj[m] == want and nobs_rec[m] == want, m
    assert cntg[0] > 0 and cntg[1] == int(((kf_fixed[:NLOC] == 0) & (kf_bad[:NLOC] == 0)).sum()) and cntg[2] == int((mp_fixed == 0).sum())
    # H: the adapter's vpMatched (as indices into vpPoints, new entries only) equals the oracle's on the flat views; entries held on entry are untouched
    mH = I32(rec[36]); cH = I32(rec[37])
    rH = pyorc.search_by_projection_scw(scH["kf2"], (heldH != -1).astype(np.uint8), SH, scH["pts1"], scH["desc1"], 10.0)
    assert np.array_equal(mH, rH[0]) and cH[0] == rH[1] and rH[1] > 30 and cH[1] == cH[2] == int((heldH != -1).sum())
    # I: SearchForInitialization through the reference's signature equals the oracle (matches, vbPrevMatched, count)
    rI = pyorc.search_for_initialization(fI1, fI2, pmI, 100, 0.9, True)
    assert np.array_equal(I32(rec[38]), rI[0]) and np.array_equal(F32(rec[39]).reshape(-1, 2), rI[1]) and int(I32(rec[40])[0]) == rI[2] and rI[2] > 100
