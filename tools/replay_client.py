#!/usr/bin/env python3
"""replay_client.py -- BASELINE configs[2] harness: one client's Tracking + LocalMapping loop and the server's global BA on ONE GPU, on a synthetic
sequence, every stage through the C-ABI and (optionally) checked against the CPU oracle on the same inputs.

Per frame (Tracking::GrabImageStereo -> Track, corbslam_client/src/Tracking.cc:166-507):
  1. stereo front-end on a synthetic 1241x376 image pair: 2 x ORBextractor::operator() + Frame::ComputeStereoMatches        (Frame.cc:61-117)
  2. TrackWithMotionModel: SearchByProjection(CurrentFrame, LastFrame, th, bMono=false) + Optimizer::PoseOptimization        (Tracking.cc:886-951)
  3. TrackLocalMap: SearchByProjection(Frame, local MapPoints, th) + PoseOptimization                                         (Tracking.cc:951-1010)
Per new keyframe (LocalMapping::Run, corbslam_client/src/LocalMapping.cc:44-108; every `kf_every`-th frame):
  4. SearchForTriangulation against the previous keyframes (CreateNewMapPoints, LocalMapping.cc:190-420) -- host pointers AND keyframe-store slots
  5. Fuse of the new keyframe's map points into its neighbours (SearchInNeighbors, :422-560)
  6. Optimizer::LocalBundleAdjustment on the window                                                                           (:79)
Every `gba_every` keyframes (the harness' stand-in for the server's fusion event, corbslam_server/src/GlobalOptimize.cpp:435-547):
  7. Optimizer::GlobalBundleAdjustemnt(cache, 10, &stop, nLoopKF, false) over all keyframes / map points

The geometry of stages 2-7 comes from a synthetic WORLD (landmarks with descriptors along a corridor, a smooth trajectory): the frames of those stages
are synthesised at feature level from the world (keypoints = noisy projections, descriptors = noisy copies), so that tracking, triangulation, fusion
and the two bundle adjustments see a consistent map; stage 1 runs on synthetic images (its keypoints are not tied to the world -- no renderer here).
The product's outputs are fed forward; the oracle runs each stage on the same inputs as a checker (check=True): matchers bit-exact, optimisers
identical outlier sets and 1e-4 estimates.  Host-side bookkeeping of the reference (KeyFrame / MapPoint graph, isInFrustum, triangulation) is numpy here.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CAM = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, w=1241, h=376)


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


class World:
    def __init__(self, seed, n_frames, step=0.8, density=260):
        rng = np.random.default_rng(seed)
        self.rng = rng
        L = n_frames * step + 50.0
        n = int(density * L)
        self.X = np.stack([rng.uniform(-18, 18, n), rng.uniform(-3.5, 3.5, n), rng.uniform(2, L, n)], 1)
        self.Xest = (self.X + rng.normal(0, 0.03, self.X.shape)).astype(np.float32)          # what triangulation would have produced
        self.desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        self.node = (self.desc[:, 0].astype(np.uint32) * 7 + self.desc[:, 1]) % 120         # vocabulary node of the landmark's descriptor
        self.angle = rng.uniform(0, 360, n).astype(np.float32)
        self.octave = rng.integers(0, 7, n)
        self.step = step
        self.scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)

    def pose(self, t):
        yaw = 0.05 * np.sin(0.07 * t)
        R = _rot_y(yaw)
        C = np.array([1.5 * np.sin(0.05 * t), 0.05 * np.sin(0.11 * t), self.step * t])
        T = np.eye(4); T[:3, :3] = R.T; T[:3, 3] = -R.T @ C
        return T

    def observe(self, t, n_feat=2000):
        """feature-level frame t: (keys KP_DTYPE, u_right, desc, landmark id per feature)"""
        from corb_slam_amd import synth
        rng = np.random.default_rng(1_000_003 * (t + 1))
        T = self.pose(t)
        Xc = (T[:3, :3] @ self.X.T).T + T[:3, 3]
        z = Xc[:, 2]
        u = CAM["fx"] * Xc[:, 0] / np.where(z > 0.1, z, 1.0) + CAM["cx"]; v = CAM["fy"] * Xc[:, 1] / np.where(z > 0.1, z, 1.0) + CAM["cy"]
        vis = np.nonzero((z > 4) & (z < 45) & (u > 8) & (u < CAM["w"] - 8) & (v > 8) & (v < CAM["h"] - 8))[0]
        if len(vis) > n_feat:
            vis = np.sort(vis[np.argsort((self.desc[vis, 2].astype(np.int64) * 131 + vis) % 9973)[:n_feat]])      # a landmark-stable subset
        m = len(vis)
        keys = np.zeros(m, synth.KP_DTYPE)
        sig = self.scale[self.octave[vis]]
        keys["x"] = u[vis] + rng.normal(0, 0.5, m) * sig; keys["y"] = v[vis] + rng.normal(0, 0.5, m) * sig
        keys["octave"] = self.octave[vis]; keys["angle"] = (self.angle[vis] + rng.normal(0, 5, m)) % 360; keys["size"] = 31 * sig; keys["class_id"] = -1
        bits = np.unpackbits(self.desc[vis], axis=1); bits ^= (rng.random(bits.shape) < 0.03).astype(np.uint8)
        desc = np.packbits(bits, axis=1)
        ur = np.where(rng.random(m) < 0.85, keys["x"] - CAM["bf"] / z[vis] + rng.normal(0, 0.4, m) * sig, -1.0).astype(np.float32)
        ur = np.where((ur < 0) & (ur > -1), 0.0, ur).astype(np.float32)
        perm = rng.permutation(m)
        return keys[perm], ur[perm], desc[perm], vis[perm]


def _frame_view(w, keys, ur, desc, claimed=None):
    return dict(keys_un=keys, u_right=ur, desc=desc, claimed=np.zeros(len(keys), np.uint8) if claimed is None else claimed,
                min_x=0.0, min_y=0.0, max_x=float(CAM["w"]), max_y=float(CAM["h"]), scale=w.scale)


def _kf_view(w, keys, ur, desc):
    f32 = lambda x: float(np.float32(x))
    return dict(keys_un=keys, u_right=ur, desc=desc, min_x=0.0, min_y=0.0, max_x=float(CAM["w"]), max_y=float(CAM["h"]), scale=w.scale,
                inv_level_sigma2=(1.0 / (w.scale * w.scale)).astype(np.float32), log_scale_factor=f32(np.log(np.float32(1.2))),
                fx=f32(CAM["fx"]), fy=f32(CAM["fy"]), cx=f32(CAM["cx"]), cy=f32(CAM["cy"]), bf=f32(CAM["bf"]))


def _feature_vector(node_of_feature):
    """DBoW2::FeatureVector of a frame whose feature i fell into vocabulary node node_of_feature[i]: ascending nodes, ascending features inside"""
    order = np.lexsort((np.arange(len(node_of_feature)), node_of_feature))
    nodes, start = np.unique(node_of_feature[order], return_index=True)
    off = np.concatenate([start, [len(order)]]).astype(np.int32)
    return nodes.astype(np.uint32), off, order.astype(np.uint32)


class _TimedOracle:
    """the oracle module with every call timed (stage totals in `acc`): the CPU baseline of the same loop, measured while it checks the product"""
    def __init__(self, mod, acc):
        self._m, self._acc = mod, acc
    def __getattr__(self, name):
        v = getattr(self._m, name)
        if not callable(v) or isinstance(v, type):
            return v
        def call(*a, **k):
            t0 = time.perf_counter(); r = v(*a, **k); self._acc[name] = self._acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        return call


class Replay:
    def __init__(self, corb, synth, pyorc=None, n_frames=24, kf_every=4, gba_every=50, seed=9000, images=True, check=True, device=0, records=False):
        self.t_cpu = {}            # oracle function -> seconds (check = True): the CPU baseline of this loop
        self.corb, self.synth, self.pyorc = corb, synth, (_TimedOracle(pyorc, self.t_cpu) if pyorc is not None else None)
        self.check = check and pyorc is not None
        self.n_stereo_checks = 0
        self.n_frames, self.kf_every, self.gba_every, self.images = n_frames, kf_every, gba_every, images
        self.w = World(seed, n_frames)
        self.f32 = {k: float(np.float32(v)) for k, v in CAM.items() if k in ("fx", "fy", "cx", "cy", "bf")}
        self.mb = float(np.float32(CAM["bf"]) / np.float32(CAM["fx"]))
        self.device = device
        self.t_gpu = {}            # stage -> seconds spent in the product's calls
        self.n_checked = {}        # stage -> oracle comparisons that passed
        self.in_map = np.zeros(len(self.w.X), bool)
        self.kfs = []              # dict(T (4x4 f32), keys, ur, desc, lm, fv, slot)
        self.store = corb.KeyFrameStore(max(8, n_frames // kf_every + 2), 2048, device=device)
        self.sf = corb.StereoFrontend(max_frames=1, device=device) if images else None
        if self.sf is not None:
            self.pin_out = corb.pinned_empty((self.sf.frame_layout().frame_bytes,), np.uint8)
            self.frame_pool = corb.pinned_empty((64, 2, 376, 1241), np.uint8)        # the 64 synthetic frames the sequence cycles through, in page-locked memory
            for i in range(64):
                l, r = synth.stereo_pair(i); self.frame_pool[i, 0] = l; self.frame_pool[i, 1] = r
        self.matcher = corb.ORBmatcher(0.9, True, device=device)
        if hasattr(corb, "warmup"):
            corb.warmup(device)    # process start-up (corb_warmup): the per-device workspace lanes
        self.errors = []
        self.stats = {}            # name -> [sum, count]
        # map-point fields Frame::isInFrustum reads (MapPoint::UpdateNormalAndDepth at creation): viewing direction, scale-invariance distances
        nL = len(self.w.X)
        self.mp_normal = np.zeros((nL, 3), np.float32); self.mp_min = np.zeros(nL, np.float32); self.mp_max = np.zeros(nL, np.float32); self.mp_init = np.zeros(nL, bool)
        self.log_scale = float(np.float32(np.log(np.float32(1.2))))
        # records = True: the tracking stages (2, 3) run on device-resident records (corb_track_*): the frames are slots of a two-slot keyframe store, the
        # map a map-point store the harness mirrors after every keyframe -- same inputs, same stage names, results equal to the host-pointer mode
        self.records = records
        if records:
            self.OBS = 64                                         # mObservations capacity of a map-point record (asserted in _observe)
            self.fstore = corb.KeyFrameStore(2, 2048, device=device); self.mstore = corb.MapPointStore(nL, self.OBS, device=device)
            self.obs_kf = np.zeros((nL, self.OBS), np.uint64); self.obs_idx = np.zeros((nL, self.OBS), np.uint32); self.obs_n = np.zeros(nL, np.int32)
            self.cam = corb.TrackCamera.make(self.f32["fx"], self.f32["fy"], self.f32["cx"], self.f32["cy"], self.f32["bf"], self.mb, 0.0, float(CAM["w"]), 0.0, float(CAM["h"]), self.w.scale)
            self.mrec = np.zeros(nL, corb.MP_RECORD_DTYPE); self.mrec["id"] = np.arange(nL); self.mrec["descriptor"] = self.w.desc
            self.fmeta = np.zeros((), corb.KF_META_DTYPE)
            for k in ("fx", "fy", "cx", "cy", "bf"):
                self.fmeta[k] = self.f32[k]
            self.fmeta["nlevels"] = 8; self.fmeta["inv_level_sigma2"][:8] = (1.0 / (self.w.scale * self.w.scale)).astype(np.float32)
            self.fmeta["Tcw"] = np.eye(4, dtype=np.float32).reshape(16); self.fmeta["TcwGBA"] = np.eye(4, dtype=np.float32).reshape(16)

    # ---- helpers ----
    def _timed(self, stage, fn, *a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); self.t_gpu[stage] = self.t_gpu.get(stage, 0.0) + time.perf_counter() - t0
        return r

    def _stat(self, name, v):
        a = self.stats.setdefault(name, [0.0, 0]); a[0] += float(v); a[1] += 1

    def _ok(self, stage, cond, what=""):
        if cond:
            self.n_checked[stage] = self.n_checked.get(stage, 0) + 1
        else:
            self.errors.append("%s: %s" % (stage, what))

    def _pose_inputs(self, lm, keys, ur):
        pts = self.w.Xest[lm]; obs = np.stack([keys["x"], keys["y"], ur], 1).astype(np.float32)
        w = (1.0 / (self.w.scale[keys["octave"]] ** 2)).astype(np.float32)
        return pts, obs, w

    def _pose_opt(self, stage, T0, lm, keys, ur):
        pts, obs, w = self._pose_inputs(lm, keys, ur)
        T, outl, ninl = self._timed(stage, self.corb.Optimizer.PoseOptimization, T0, pts, obs, w, self.f32["fx"], self.f32["fy"], self.f32["cx"], self.f32["cy"], self.f32["bf"], device=self.device)
        self._check_pose(stage, T0, lm, keys, ur, T, outl)
        return T, outl

    def _check_pose(self, stage, T0, lm, keys, ur, T, outl):
        """the oracle's PoseOptimization on the same edges against the product's result (either mode)"""
        if not self.check:
            return
        pts, obs, w = self._pose_inputs(lm, keys, ur)
        n = len(pts); e = np.zeros(n, self.pyorc.EDGE_DTYPE)
        e["pose"] = 0; e["point"] = np.arange(n); e["u"] = obs[:, 0]; e["v"] = obs[:, 1]; e["ur"] = obs[:, 2]; e["inv_sigma2"] = w
        st = self.pyorc.POSE_OPT_STAGES if n >= 10 else self.pyorc.POSE_OPT_STAGES[:1]
        r = self.pyorc.ba_solve_staged(np.asarray(T0, np.float32).reshape(1, 16), np.zeros(1, np.uint8), pts, np.ones(n, np.uint8), e,
                                       self.f32["fx"], self.f32["fy"], self.f32["cx"], self.f32["cy"], self.f32["bf"], st)
        self._ok(stage, np.array_equal(outl, r["outlier"].astype(bool)) and np.abs(np.asarray(T).reshape(4, 4) - r["poses"][0].reshape(4, 4)).max() <= 1e-4 * max(1.0, np.abs(r["poses"][0]).max()), "pose / outliers differ from the oracle")

    # ---- the loop ----
    def run(self):
        corb, w = self.corb, self.w
        last = None; velocity = np.eye(4)
        for t in range(self.n_frames):
            # 1. stereo front-end on an image pair
            if self.images:
                l, r = self.frame_pool[t % 64, 0], self.frame_pool[t % 64, 1]
                def front():
                    # corb_stereo_frames: the client's per-frame call.  The camera driver's buffers are page-locked (the pool below stands for them): the
                    # transfer reads the frame where it arrived, no staging copy
                    return self.sf.unpack_frame(self.sf.frames(self.frame_pool[t % 64: t % 64 + 1], self.pin_out))
                out = self._timed("1 stereo front-end", front)
                if self.check and t % 6 == 0:
                    el, er = self.pyorc.Extractor(), self.pyorc.Extractor()
                    t0 = time.perf_counter()
                    kl, dl = el.extract(l); kr, dr = er.extract(r); tb = el.tables()
                    self.t_cpu["extract x2"] = self.t_cpu.get("extract x2", 0.0) + time.perf_counter() - t0; self.n_stereo_checks += 1
                    ur_o, dp_o, nm = self.pyorc.stereo_match(el, er, kl, dl, kr, dr, 386.1448, 718.856, tb["scale"], tb["inv_scale"])
                    self._ok("1 stereo front-end", out["kl"].tobytes() == kl.tobytes() and np.array_equal(out["dl"], dl) and np.array_equal(out["u_right"].view(np.uint32), ur_o.view(np.uint32)), "extraction / stereo match differ")
            keys, ur, desc, lm = w.observe(t)
            fv = _frame_view(w, keys, ur, desc)
            if last is None:                                     # StereoInitialization: first keyframe at the true pose, its stereo points enter the map
                T = w.pose(0).astype(np.float32)
                self._new_keyframe(t, T, keys, ur, desc, lm)
                last = dict(T=T, keys=keys, ur=ur, desc=desc, lm=lm, outlier=np.zeros(len(keys), bool))
                if self.records:
                    self.fmeta["id"] = 1; self.fstore.put_frame(0, keys, desc, ur, None, self.fmeta)
                    self._frame_to_record(0, last)
                continue
            # 2. TrackWithMotionModel (Tracking.cc:868-940): SearchByProjection(CurrentFrame, LastFrame, 7) -> PoseOptimization -> discard outliers
            T_pred = (velocity @ last["T"].astype(np.float64)).astype(np.float32)
            lastp = np.zeros(len(last["lm"]), corb.LAST_DTYPE)
            lastp["world"] = w.Xest[last["lm"]]; lastp["angle"] = last["keys"]["angle"]; lastp["octave"] = last["keys"]["octave"]
            lastp["valid"] = self.in_map[last["lm"]] & ~last["outlier"]; lastp["claims"] = 1
            ldesc = w.desc[last["lm"]]                            # pMP->GetDescriptor(): the landmark's representative descriptor
            if self.records:
                slot = t & 1
                def search_last():
                    self.fmeta["id"] = t + 1; self.fstore.put_frame(slot, keys, desc, ur, None, self.fmeta)     # (the frame reaches the device once: counted with its first use)
                    return self.fstore.TrackSearchLastFrame(slot, 1 - slot, self.mstore, T_pred, last["T"], self.cam, 7.0, mono=False, nnratio=0.9, check_orientation=True)
                m, n = self._timed("2 SearchByProjection(frame,last)", search_last)
            else:
                a = (fv, T_pred, last["T"], self.f32["fx"], self.f32["fy"], self.f32["cx"], self.f32["cy"], self.f32["bf"], self.mb, lastp, ldesc, 7.0, False)
                m, n = self._timed("2 SearchByProjection(frame,last)", self.matcher.SearchByProjection_Frame, *a)
            if self.check:
                mo, no = self.pyorc.search_by_projection_frame(fv, T_pred, last["T"], self.f32["fx"], self.f32["fy"], self.f32["cx"], self.f32["cy"], self.f32["bf"], self.mb, lastp, ldesc, 7.0, 0, 1)
                self._ok("2 SearchByProjection(frame,last)", n == no and np.array_equal(m, mo), "matches differ")
            got = m >= 0; self._stat("matches to the last frame", n)
            f_lm = np.full(len(keys), -1); f_lm[got] = last["lm"][m[got]]
            if self.records:
                T, ofull, _ = self._timed("2 PoseOptimization", self.fstore.TrackPoseOptimization, slot, self.mstore, self.cam, T_pred, True)
                outl = ofull[got]
                self._check_pose("2 PoseOptimization", T_pred, f_lm[got], keys[got], ur[got], T, outl)
            else:
                T, outl = self._pose_opt("2 PoseOptimization", T_pred, f_lm[got], keys[got], ur[got])
            disc = np.zeros(len(keys), bool); disc[np.nonzero(got)[0][outl]] = True      # discarded: no MapPoint any more, but seen in this frame (mnLastFrameSeen)
            seen = f_lm[got]
            f_lm[disc] = -1
            # 3. TrackLocalMap (Tracking.cc:1040-1090): SearchLocalPoints = isInFrustum of the local map points the frame has not seen + SearchByProjection(F, points, 1)
            local_all = np.unique(np.concatenate([k["lm"][self.in_map[k["lm"]]] for k in self.kfs[-6:]]))
            local = local_all[~np.isin(local_all, seen)]
            mps = self.synth.frustum_view(T, w.Xest[local], self.mp_normal[local], self.mp_min[local], self.mp_max[local], self.f32["fx"], self.f32["fy"], self.f32["cx"], self.f32["cy"],
                                          self.f32["bf"], 0.0, float(CAM["w"]), 0.0, float(CAM["h"]), self.log_scale, 8)
            mps = mps.astype(corb.TRACKED_DTYPE); mps["claims"] = mps["valid"]
            ldesc3 = np.where(mps["valid"][:, None].astype(bool), w.desc[local], 0).astype(np.uint8)
            fv3 = _frame_view(w, keys, ur, desc, claimed=(f_lm >= 0).astype(np.uint8))
            if self.records:
                mr, n3, _ = self._timed("3 SearchByProjection(frame,map)", self.fstore.TrackSearchLocalPoints, slot, self.mstore, local_all, self.cam, T, self.log_scale, 1.0, 0.9)
                pos = np.searchsorted(local, local_all[np.maximum(mr, 0)])               # index into local_all -> index into the caller-side list `local`
                m3 = np.where(mr >= 0, pos, -1).astype(np.int32)
            else:
                m3, n3 = self._timed("3 SearchByProjection(frame,map)", self.matcher.SearchByProjection, fv3, mps, ldesc3, 1.0)
            if self.check:
                mo, no = self.pyorc.search_by_projection_map(fv3, mps, ldesc3, 1.0, 0.9)
                self._ok("3 SearchByProjection(frame,map)", n3 == no and np.array_equal(m3, mo), "matches differ")
            new = (m3 >= 0) & (f_lm < 0); self._stat("extra matches to the local map", int(new.sum()))
            f_lm[new] = local[m3[new]]
            have = f_lm >= 0
            if self.records:
                T3, ofull, _ = self._timed("3 PoseOptimization", self.fstore.TrackPoseOptimization, slot, self.mstore, self.cam, T, False)
                outl = ofull[have]
                self._check_pose("3 PoseOptimization", T, f_lm[have], keys[have], ur[have], T3, outl)
                T = T3
            else:
                T, outl = self._pose_opt("3 PoseOptimization", T, f_lm[have], keys[have], ur[have])
            outlier = disc.copy(); outlier[np.nonzero(have)[0][outl]] = True              # (the harness keeps both kinds out of the next frame's search)
            velocity = T.astype(np.float64) @ np.linalg.inv(last["T"].astype(np.float64))
            last = dict(T=T, keys=keys, ur=ur, desc=desc, lm=np.where(f_lm >= 0, f_lm, lm), outlier=outlier)   # (unmatched features keep their true landmark for the next KF)
            self.track_err = float(np.abs(T[:3, 3] - w.pose(t)[:3, 3]).max())
            if t % self.kf_every == 0:
                self._new_keyframe(t, T, keys, ur, desc, lm)
            if self.records:
                self._frame_to_record(slot, last)           # (after the keyframe: in_map is the one the next frame's search sees)
        return self.report()

    def _frame_to_record(self, slot, fr):
        """the harness's convention for the NEXT frame's search (every feature whose landmark is in the map holds it; outliers flagged) written into the record"""
        NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
        self.fstore.set_map_points(slot, np.where(self.in_map[fr["lm"]], fr["lm"].astype(np.uint64), NONE))
        self.fstore.set_flags(slot, np.where(fr["outlier"], 2, 0).astype(np.uint8))

    def _sync_map(self, kf_T):
        """MapPoint::UpdateNormalAndDepth for the landmarks that just entered the map (viewing direction and distance range from the creating keyframe); the
        map-point store mirrors the harness's map (records mode)"""
        w = self.w
        fresh = np.nonzero(self.in_map & ~self.mp_init)[0]
        if len(fresh):
            T = np.asarray(kf_T, np.float64)
            Ow = -T[:3, :3].T @ T[:3, 3]
            PO = w.Xest[fresh].astype(np.float64) - Ow; dist = np.linalg.norm(PO, axis=1)
            self.mp_normal[fresh] = (PO / dist[:, None]).astype(np.float32)
            self.mp_max[fresh] = (dist * w.scale[w.octave[fresh]]).astype(np.float32); self.mp_min[fresh] = (self.mp_max[fresh] / w.scale[7]).astype(np.float32)
            self.mp_init[fresh] = True
        if self.records:
            self._upload_map()

    def _observe(self, kf_id, lm):
        """mObservations of the landmarks a new keyframe sees (MapPoint::AddObservation): the harness's mirror of the lists the records carry"""
        n = self.obs_n[lm]
        assert n.max() < self.OBS, "a landmark has more than %d observations" % self.OBS
        self.obs_kf[lm, n] = kf_id; self.obs_idx[lm, n] = np.arange(len(lm)); self.obs_n[lm] = n + 1

    def _upload_map(self):
        """the map-point store mirrors the harness's map (records mode): headers + mObservations of the landmarks in the map"""
        r = self.mrec
        r["flags"] = np.where(self.in_map, 0, 1); r["n_obs"] = np.where(self.in_map, self.obs_n, 0); r["world_pos"] = self.w.Xest
        r["normal"] = self.mp_normal; r["min_distance"] = self.mp_min; r["max_distance"] = self.mp_max; r["ref_kf_id"] = self.obs_kf[:, 0]
        off = np.concatenate([[0], np.cumsum(r["n_obs"])]).astype(np.int32)
        sel = np.arange(self.OBS)[None, :] < r["n_obs"][:, None]
        self.mstore.put(0, r, off, self.obs_kf[sel], self.obs_idx[sel])
        self.mstore.build_index(0, len(r))

    def _new_keyframe(self, t, T, keys, ur, desc, lm):
        corb, w = self.corb, self.w
        k = dict(t=t, T=np.asarray(T, np.float32), keys=keys, ur=ur, desc=desc, lm=lm, fv=_feature_vector(w.node[lm]), slot=len(self.kfs))
        before = self.in_map.copy()
        has_mp = self.in_map[lm].astype(np.uint8)
        self.store.put(k["slot"], keys, desc, ur, None, keyframe_id=len(self.kfs) + 1); self.store.set_bow(k["slot"], k["fv"]); self.store.set_flags(k["slot"], has_mp)
        if self.records:                                         # the keyframe's header (pose, intrinsics, mvInvLevelSigma2) and its landmarks' mObservations: what the bundle adjustments read
            self.fmeta["id"] = k["slot"] + 1; self.fmeta["Tcw"] = k["T"].reshape(16); self.store.set_meta_raw(k["slot"], self.fmeta)
            self._observe(k["slot"] + 1, lm)
        sigma2 = (w.scale * w.scale).astype(np.float32)
        # 4. CreateNewMapPoints: SearchForTriangulation against the previous keyframes (the fundamental matrix from the two poses, LocalMapping::ComputeF12)
        for prev in self.kfs[-3:]:
            T1, T2 = k["T"].astype(np.float64), prev["T"].astype(np.float64)
            R12 = T1[:3, :3] @ T2[:3, :3].T; t12 = -R12 @ T2[:3, 3] + T1[:3, 3]
            K = np.array([[CAM["fx"], 0, CAM["cx"]], [0, CAM["fy"], CAM["cy"]], [0, 0, 1.0]])
            tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
            F12 = (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype(np.float32)
            Ow1 = -T1[:3, :3].T @ T1[:3, 3]; C2 = T2[:3, :3] @ Ow1 + T2[:3, 3]
            ex = float(np.float32(CAM["fx"] * C2[0] / C2[2] + CAM["cx"])); ey = float(np.float32(CAM["fy"] * C2[1] / C2[2] + CAM["cy"]))
            pm = self.in_map[prev["lm"]].astype(np.uint8)
            self.store.set_flags(prev["slot"], pm)
            A = dict(desc=desc, kp=keys, u_right=ur, has_mp=has_mp, fv=k["fv"]); B = dict(desc=prev["desc"], kp=prev["keys"], u_right=prev["ur"], has_mp=pm, fv=prev["fv"])
            m0 = corb.ORBmatcher(0.6, True, device=self.device)
            # both forms (host pointers, store slots) when the run is checked; an unchecked (timed) run uses the form of its mode only -- round 3's client_loop figures
            # carried both calls in both modes, i.e. 45 - 75 ms per 400 frames of a second, redundant search
            if self.check or not self.records:
                pairs, n = self._timed("4 SearchForTriangulation", m0.SearchForTriangulation, A, B, F12, ex, ey, w.scale, sigma2, False)
            if self.check or self.records:
                ps, ns = self._timed("4 SearchForTriangulation (store slots)", self.store.SearchForTriangulation, k["slot"], self.store, prev["slot"], F12, ex, ey, w.scale, sigma2, False)
                if self.check or not self.records:
                    self._ok("4 SearchForTriangulation (store slots)", ns == n and np.array_equal(ps, pairs), "slot call differs from the host-pointer call")
                else:
                    pairs, n = ps, ns
            if self.check:
                rp, rn = self.pyorc.search_for_triangulation(desc, keys, ur, has_mp, self.pyorc.FeatVec(*k["fv"]), prev["desc"], prev["keys"], prev["ur"], pm, self.pyorc.FeatVec(*prev["fv"]),
                                                             F12, ex, ey, w.scale, sigma2, False, True)
                self._ok("4 SearchForTriangulation", n == rn and np.array_equal(pairs.reshape(-1, 2), np.asarray(rp).reshape(-1, 2)), "pairs differ")
            good = pairs[lm[pairs[:, 0]] == prev["lm"][pairs[:, 1]]] if n else pairs
            self._stat("triangulation pairs", n); self._stat("triangulation pairs with the right landmark", len(good))
            self.in_map[lm[good[:, 0]]] = True                   # triangulated (host-side arithmetic of the reference; the estimate is w.Xest)
            has_mp = self.in_map[lm].astype(np.uint8)
        self.in_map[lm[ur >= 0]] = True                          # close stereo points (Tracking::CreateNewKeyFrame, Tracking.cc:1126-1190)
        self.kfs.append(k)
        # 5. SearchInNeighbors: Fuse the new keyframe's map points into the previous keyframe
        if len(self.kfs) >= 2:
            prev = self.kfs[-2]
            mine = np.unique(lm[self.in_map[lm] & ~before[lm]])             # the map points this keyframe just created: the neighbour may see them too
            pts = np.zeros(len(mine), corb.MP_DTYPE)
            Op = -prev["T"][:3, :3].astype(np.float64).T @ prev["T"][:3, 3]
            PO = w.Xest[mine] - Op; dist = np.linalg.norm(PO, axis=1)
            pts["world"] = w.Xest[mine]; pts["normal"] = (PO / dist[:, None]).astype(np.float32)
            pts["max_distance"] = (dist * w.scale[w.octave[mine]]).astype(np.float32) * 1.2; pts["min_distance"] = pts["max_distance"] / w.scale[7] / 1.44
            pts["valid"] = 1                                                       # (none of them is in the neighbour yet: !pMP->IsInKeyFrame(pKF))
            kv = _kf_view(w, prev["keys"], prev["ur"], prev["desc"])
            bi, bd, nf = self._timed("5 Fuse", self.matcher.Fuse, kv, prev["T"], Op.astype(np.float32), pts, w.desc[mine], 3.0)
            self._stat("fused points", nf)
            if self.check:
                ro = self.pyorc.fuse(kv, prev["T"], Op.astype(np.float32), 0, pts, w.desc[mine], 3.0)
                self._ok("5 Fuse", nf == ro[-1] and np.array_equal(bi, ro[0]), "fused features differ")
        # 6. LocalBundleAdjustment on the window; 7. global BA every gba_every keyframes
        if len(self.kfs) >= 3:
            self._bundle("6 LocalBundleAdjustment", self.kfs[-5:], self.kfs[-9:-5], local=True)
        if len(self.kfs) % self.gba_every == 0:
            self._bundle("7 GlobalBundleAdjustemnt", self.kfs, [], local=False)
        self._sync_map(k["T"])

    def _bundle(self, stage, free_kfs, fixed_kfs, local):
        corb, w = self.corb, self.w
        kfs = free_kfs + fixed_kfs
        pts_id = np.unique(np.concatenate([k["lm"][self.in_map[k["lm"]]] for k in free_kfs]))
        idx = -np.ones(len(w.X), np.int64); idx[pts_id] = np.arange(len(pts_id))
        E = []
        for j, k in enumerate(kfs):
            sel = np.nonzero(idx[k["lm"]] >= 0)[0]
            e = np.zeros(len(sel), self.synth.EDGE_DTYPE)
            e["pose"] = j; e["point"] = idx[k["lm"][sel]]; e["u"] = k["keys"]["x"][sel]; e["v"] = k["keys"]["y"][sel]; e["ur"] = k["ur"][sel]
            e["inv_sigma2"] = 1.0 / (w.scale[k["keys"]["octave"][sel]] ** 2)
            E.append(e)
        E = np.concatenate(E)
        # edges per map point, within a point in ascending keyframe id: the order the reference creates them in (lLocalMapPoints x mObservations,
        # Optimizer.cc:626-700) and the order the records yield -- both modes solve the same lists
        E = E[np.lexsort((np.array([k["slot"] for k in kfs])[E["pose"]], E["point"]))]
        poses = np.stack([k["T"] for k in kfs]).astype(np.float32)
        fixed = np.zeros(len(kfs), np.uint8); fixed[len(free_kfs):] = 1
        for j, k in enumerate(kfs):
            if k["slot"] == 0:
                fixed[j] = 1                                     # mnId == 1 (Optimizer.cc:94, :553)
        a = (poses, fixed, w.Xest[pts_id], np.zeros(len(pts_id), np.uint8), E, self.f32["fx"], self.f32["fy"], self.f32["cx"], self.f32["cy"], self.f32["bf"])
        slots = np.array([k["slot"] for k in kfs], np.int32)
        if self.records:
            self._upload_map()                                   # (mirror: the landmarks this keyframe brought into the map become records)
        if local:
            if self.records:
                # on records: graph from the keyframe / map-point records on the device; the harness keeps every observation, like the host-pointer mode
                # (which drops g["outlier"]), so that both modes stay the same map: apply_erase = False
                g = self._timed(stage, corb.LocalBundleAdjustmentStore, self.store, slots, len(free_kfs), self.mstore, pts_id, float(w.scale[1]), False)
                pair = {(int(p), int(q)): i for i, (p, q) in enumerate(zip(E["pose"], E["point"]))}
                g["outlier"] = np.zeros(len(E), np.uint8); g["outlier"][[pair[(int(p), int(q))] for p, q in g["erase"]]] = 1
            else:
                g = self._timed(stage, corb.Optimizer.LocalBundleAdjustment, *a, device=self.device)
            if self.check:
                r = self.pyorc.ba_solve_staged(*a, self.pyorc.LOCAL_BA_STAGES)
                self._ok(stage, np.array_equal(g["outlier"], r["outlier"]) and np.abs(g["poses"] - r["poses"]).max() <= 1e-4 * max(1.0, np.abs(r["poses"]).max()) and
                         np.abs(g["points"] - r["points"]).max() <= 1e-4 * max(1.0, np.abs(r["points"]).max()), "estimates / vToErase differ")
        else:
            if self.records:
                g = self._timed(stage, corb.GlobalBundleAdjustemntStore, self.store, slots, self.mstore, pts_id, nIterations=10, bRobust=False)
            else:
                g = self._timed(stage, corb.Optimizer.GlobalBundleAdjustemnt, *a, nIterations=10, bRobust=False, device=self.device)
            if self.check:
                r = self.pyorc.ba_solve(*a, iters=10, robust=False)
                self._ok(stage, g["iters_done"] == r["iters_done"] and np.allclose(g["chi2"], r["chi2"], rtol=1e-4) and np.abs(g["poses"] - r["poses"]).max() <= 1e-4 * max(1.0, np.abs(r["poses"]).max()),
                         "chi2 / poses differ")
        for j, k in enumerate(kfs):
            if not fixed[j]:
                k["T"] = g["poses"][j].astype(np.float32)
        w.Xest[pts_id] = g["points"]

    def cpu_baseline(self):
        """the oracle's time for the SAME calls (one thread; the stereo front-end is checked on every 6th frame only: scaled to every frame)"""
        if not self.check or not self.t_cpu:
            return None
        scale = self.n_frames / max(self.n_stereo_checks, 1)
        front = (self.t_cpu.get("extract x2", 0.0) + self.t_cpu.get("stereo_match", 0.0)) * scale
        rest = sum(v for k, v in self.t_cpu.items() if k not in ("extract x2", "stereo_match", "Extractor"))
        total = front + rest
        return dict(value=round(self.n_frames / total, 2) if total > 0 else None, unit="client frames/s", cores=1, kind="port",
                    sample="the oracle on the same %d frames / %d keyframes, every stage on the inputs the product saw (stereo front-end timed on every 6th frame and scaled)" % (self.n_frames, len(self.kfs)),
                    stage_ms=dict([("stereo front-end", round(front * 1e3, 1))] + [(k, round(v * 1e3, 1)) for k, v in sorted(self.t_cpu.items()) if k not in ("extract x2", "stereo_match", "Extractor")]))

    def report(self):
        total = sum(self.t_gpu.values())
        cpu = self.cpu_baseline()
        return dict(**({"cpu_baseline": cpu} if cpu else {}), **self._report(total))

    def _report(self, total):
        return dict(frames=self.n_frames, keyframes=len(self.kfs), map_points=int(self.in_map.sum()), client_fps=round(self.n_frames / total, 1) if total > 0 else None,
                    mean={k: round(v[0] / max(v[1], 1), 1) for k, v in sorted(self.stats.items())}, stage_ms={k: round(v * 1e3, 2) for k, v in sorted(self.t_gpu.items())}, checks_passed=dict(sorted(self.n_checked.items())), errors=self.errors,
                    final_tracking_error_m=round(getattr(self, "track_err", 0.0), 4))

    def close(self):
        self.store.close()
        if self.records:
            self.fstore.close(); self.mstore.close()
        if self.sf is not None:
            self.sf.close()


if __name__ == "__main__":
    import json
    import corbload
    corb = corbload.load_pkg()
    from corb_slam_amd import synth
    check = "--check" in sys.argv
    pyorc = None
    if check:
        from oracle import pyorc
    n = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 200
    r = Replay(corb, synth, pyorc, n_frames=n, kf_every=4, gba_every=50 if n >= 200 else 4, check=check, records="--records" in sys.argv)
    print(json.dumps(r.run()))
    r.close()
