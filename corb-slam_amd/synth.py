"""Deterministic synthetic workloads for the CORB-SLAM hot path.

Pure numpy (PCG64 streams), so the same arrays are produced in the build container and on the GPU
box.  Used by tests/, bench.py and __graft_entry__.smoke(); not part of the product path.
"""
import numpy as np

SEED0 = 0xC02B5EED


def _value_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw), dtype=np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def stereo_pair(idx, w=1241, h=376, n_rect=300, contrast=1.0):
    """KITTI-shaped synthetic stereo pair: value-noise texture + rectangles; the right image is the
    left one shifted by a per-row-band disparity in [4,120] px plus iid +-2 noise."""
    rng = np.random.default_rng(SEED0 + int(idx))
    W = w + 128
    tex = (_value_noise(rng, h, W, 64) - 0.5) * 80.0 + (_value_noise(rng, h, W, 16) - 0.5) * 64.0
    # fine texture, amplitude modulated by a slow mask so that some FAST cells need the minTh fallback
    mask = 0.5 + 0.5 * _value_noise(rng, h, W, 96)
    tex += ((_value_noise(rng, h, W, 6) - 0.5) * 56.0 + (_value_noise(rng, h, W, 3) - 0.5) * 52.0) * mask * contrast
    img = 128.0 + tex
    ww = w + 128
    scale = max(1.0, (w * h) / (1241.0 * 376.0))
    for _ in range(int(n_rect * scale)):
        rw_, rh_ = int(rng.integers(6, 60)), int(rng.integers(6, 40))
        x = int(rng.integers(0, ww - rw_)); y = int(rng.integers(0, h - rh_))
        img[y:y + rh_, x:x + rw_] += float(rng.integers(-70, 71))
    img = np.clip(img, 0, 255)
    wide = np.rint(img).astype(np.uint8)
    left = np.ascontiguousarray(wide[:, :w])
    # piecewise-constant disparity per row band
    right = np.empty_like(left)
    y = 0
    while y < h:
        band = int(rng.integers(16, 64))
        d = int(rng.integers(4, 121))
        y1 = min(h, y + band)
        right[y:y1] = wide[y:y1, d:d + w]          # right(x) = scene(x + d)  =>  uL - uR = d
        y = y1
    noise = rng.integers(-2, 3, size=right.shape, dtype=np.int16)
    right = np.clip(right.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return left, right


def flat_image(w, h, value=128):
    return np.full((h, w), value, np.uint8)


def feature_vector(n, n_nodes, rng, drop=0.05):
    """A flat DBoW2::FeatureVector: each feature belongs to exactly one vocabulary node.
    Returns (node_id u32 ascending, offset i32, idx u32) with a fraction of features unassigned."""
    node = rng.integers(0, n_nodes, size=n)
    keep = rng.random(n) >= drop
    ids = np.unique(node[keep])
    offset = [0]; idx = []
    for nid in ids:
        m = np.nonzero((node == nid) & keep)[0]
        idx.append(m); offset.append(offset[-1] + len(m))
    idx = np.concatenate(idx) if idx else np.zeros(0, np.int64)
    # vocabulary ids are sparse in practice: spread them
    return (ids.astype(np.uint32) * 7 + 3), np.asarray(offset, np.int32), idx.astype(np.uint32)


def correlated_descriptors(n, rng, base=None, flip=0.06):
    """Random 256-bit descriptors; with `base`, a noisy permuted copy (so matches exist)."""
    if base is None:
        return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    m = len(base)
    src = rng.integers(0, m, size=n)
    bits = np.unpackbits(base[src], axis=1)
    fl = rng.random(bits.shape) < flip
    bits = bits ^ fl.astype(np.uint8)
    fresh = rng.random(n) < 0.3
    out = np.packbits(bits, axis=1)
    out[fresh] = rng.integers(0, 256, size=(int(fresh.sum()), 32), dtype=np.uint8)
    return out, src


# ------------------------------------------------------------------------------------------------
# Bundle-adjustment problems

def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


EDGE_DTYPE = np.dtype([("pose", "<i4"), ("point", "<i4"), ("u", "<f4"), ("v", "<f4"),
                       ("ur", "<f4"), ("inv_sigma2", "<f4")])


def ba_problem(n_clients=1, kf_per_client=8, pts_per_kf=12, seed=1000, fx=718.856, fy=718.856, cx=607.1928,
               cy=185.2157, bf=386.1448, w=1241, h=376, mono_frac=0.15, window=6, shared_frac=0.02,
               pose_noise=(0.02, 0.0035), point_noise=0.05, pix_noise=1.0, max_obs=8):
    """Fused multi-client stereo BA problem.  Each client drives a closed loop (1 m spacing); points lie
    in a corridor around the paths; each point is observed by up to `max_obs` nearby keyframes of its
    client (plus, for a `shared_frac` of points, keyframes of the next client).  Exactly one pose
    (index 0, the reference's mnId==1) is fixed.  Returns float32 arrays in the C-ABI layout."""
    rng = np.random.default_rng(seed)
    K = n_clients * kf_per_client
    Tcw_true = np.zeros((K, 4, 4)); centers = np.zeros((K, 3))
    k = 0
    for c in range(n_clients):
        R0 = max(kf_per_client / (2 * np.pi), 2.0)          # ~1 m spacing along a circle
        off = np.array([c * 1.5 * R0, 0.0, 0.0])
        for i in range(kf_per_client):
            th = 2 * np.pi * i / kf_per_client
            C = off + np.array([R0 * np.cos(th), 0.3 * np.sin(3 * th), R0 * np.sin(th)])
            # camera looks along the tangent (z forward), y down
            fwd = np.array([-np.sin(th), 0.0, np.cos(th)]); down = np.array([0.0, 1.0, 0.0])
            right = np.cross(down, fwd); right /= np.linalg.norm(right)
            Rwc = np.stack([right, down, fwd], axis=1)
            Rcw = Rwc.T
            T = np.eye(4); T[:3, :3] = Rcw; T[:3, 3] = -Rcw @ C
            Tcw_true[k] = T; centers[k] = C; k += 1
    sigma_oct = 1.2 ** np.arange(8)
    quota = np.array([434, 362, 302, 251, 209, 175, 145, 122], float); quota /= quota.sum()
    pts = []; edges = []
    M = K * pts_per_kf
    for m in range(M):
        kf = m // pts_per_kf
        c = kf // kf_per_client
        # a point 4-30 m in front of its anchor keyframe
        T = Tcw_true[kf]
        z = rng.uniform(4.0, 30.0)
        u = rng.uniform(40, w - 40); v = rng.uniform(30, h - 30)
        Xc = np.array([(u - cx) * z / fx, (v - cy) * z / fy, z])
        Xw = T[:3, :3].T @ (Xc - T[:3, 3])
        cand = [c * kf_per_client + ((kf - c * kf_per_client + d) % kf_per_client) for d in range(-window, window + 1)]
        if n_clients > 1 and rng.random() < shared_frac:
            c2 = (c + 1) % n_clients
            cand += [c2 * kf_per_client + int(j) for j in rng.integers(0, kf_per_client, size=3)]
        obs = []
        for j in cand:
            Tj = Tcw_true[j]
            Xj = Tj[:3, :3] @ Xw + Tj[:3, 3]
            if Xj[2] < 1.0:
                continue
            uu = fx * Xj[0] / Xj[2] + cx; vv = fy * Xj[1] / Xj[2] + cy
            if 0 <= uu < w and 0 <= vv < h:
                obs.append((j, uu, vv, Xj[2]))
        if len(obs) < 2:
            # always keep at least the anchor + one neighbour (project even if out of frame)
            obs = []
            for j in (kf, c * kf_per_client + ((kf - c * kf_per_client + 1) % kf_per_client)):
                Tj = Tcw_true[j]; Xj = Tj[:3, :3] @ Xw + Tj[:3, 3]
                if Xj[2] > 0.5:
                    obs.append((j, fx * Xj[0] / Xj[2] + cx, fy * Xj[1] / Xj[2] + cy, Xj[2]))
        if len(obs) > max_obs:
            sel = rng.choice(len(obs), size=max_obs, replace=False); obs = [obs[i] for i in sorted(sel)]
        pid = len(pts); pts.append(Xw)
        for (j, uu, vv, zz) in obs:
            octv = int(rng.choice(8, p=quota))
            s = sigma_oct[octv] * pix_noise
            un = uu + rng.normal(0, s); vn = vv + rng.normal(0, s)
            if rng.random() < mono_frac:
                ur = -1.0
            else:
                ur = un - bf / zz + rng.normal(0, s * 0.5)
            edges.append((j, pid, un, vn, ur, 1.0 / (sigma_oct[octv] ** 2)))
    pts = np.asarray(pts)
    # perturb initial estimates
    poses0 = np.zeros((K, 4, 4), np.float32)
    for k in range(K):
        T = Tcw_true[k].copy()
        if k != 0:
            dR = _rot(*(rng.normal(0, pose_noise[1], 3)))
            T[:3, :3] = dR @ T[:3, :3]; T[:3, 3] = dR @ T[:3, 3] + rng.normal(0, pose_noise[0], 3)
        poses0[k] = T.astype(np.float32)
    points0 = (pts + rng.normal(0, point_noise, pts.shape)).astype(np.float32)
    e = np.zeros(len(edges), EDGE_DTYPE)
    for i, (j, pid, un, vn, ur, isg) in enumerate(edges):
        e[i] = (j, pid, un, vn, ur, isg)
    pose_fixed = np.zeros(K, np.uint8); pose_fixed[0] = 1
    point_fixed = np.zeros(len(pts), np.uint8)
    return dict(poses=poses0, pose_fixed=pose_fixed, points=points0, point_fixed=point_fixed, edges=e,
                fx=float(np.float32(fx)), fy=float(np.float32(fy)), cx=float(np.float32(cx)), cy=float(np.float32(cy)),
                bf=float(np.float32(bf)), poses_true=Tcw_true, points_true=pts)


def local_ba_problem(seed=2000, n_local=6, n_fixed=4, pts_per_kf=25, outlier_frac=0.08, **kw):
    """Optimizer::LocalBundleAdjustment-shaped problem: `n_local` free keyframes (index 0 plays mnId==1 and is fixed like
    in the reference), `n_fixed` fixed keyframes that only observe the local points, gross outlier observations mixed in."""
    rng = np.random.default_rng(seed)
    prob = ba_problem(n_clients=1, kf_per_client=n_local + n_fixed, pts_per_kf=pts_per_kf, seed=seed, window=4, **kw)
    prob["pose_fixed"][:] = 0
    prob["pose_fixed"][0] = 1
    prob["pose_fixed"][n_local:] = 1
    # fixed keyframes keep their true pose (they are not optimised)
    prob["poses"][n_local:] = prob["poses_true"][n_local:].astype(np.float32)
    e = prob["edges"]
    bad = rng.random(len(e)) < outlier_frac
    e["u"][bad] += rng.choice([-1, 1], bad.sum()) * rng.uniform(15, 60, bad.sum())
    e["v"][bad] += rng.choice([-1, 1], bad.sum()) * rng.uniform(15, 60, bad.sum())
    prob["outlier_truth"] = bad
    return prob


def pose_opt_problem(seed=3000, n=400, outlier_frac=0.15, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448):
    """Optimizer::PoseOptimization-shaped problem: one frame pose, n fixed map points, stereo/mono observations with outliers."""
    rng = np.random.default_rng(seed)
    T = np.eye(4); T[:3, :3] = _rot(0.02, -0.03, 0.01); T[:3, 3] = [0.3, -0.1, 0.5]
    z = rng.uniform(4, 40, n); u = rng.uniform(30, 1210, n); v = rng.uniform(20, 356, n)
    Xc = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1)
    Xw = (T[:3, :3].T @ (Xc - T[:3, 3]).T).T
    octv = rng.integers(0, 8, n); sig = 1.2 ** octv
    obs = np.stack([u + rng.normal(0, 0.7, n) * sig, v + rng.normal(0, 0.7, n) * sig, u - bf / z + rng.normal(0, 0.7, n) * sig], 1)
    mono = rng.random(n) < 0.2
    obs[mono, 2] = -1.0
    bad = rng.random(n) < outlier_frac
    obs[bad, 0] += rng.choice([-1, 1], bad.sum()) * rng.uniform(10, 80, bad.sum())
    T0 = T.copy(); T0[:3, :3] = _rot(0.01, 0.01, -0.01) @ T0[:3, :3]; T0[:3, 3] += [0.05, -0.04, 0.06]
    return dict(Tcw0=T0.astype(np.float32), Tcw_true=T, points=Xw.astype(np.float32), obs=obs.astype(np.float32),
                inv_sigma2=(1.0 / sig ** 2).astype(np.float32), outlier_truth=bad,
                fx=float(np.float32(fx)), fy=float(np.float32(fy)), cx=float(np.float32(cx)), cy=float(np.float32(cy)), bf=float(np.float32(bf)))


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
TRACKED_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"), ("level", "<i4"),
                          ("valid", "u1"), ("claims", "u1"), ("pad", "u1", 2)])
LAST_DTYPE = np.dtype([("world", "<f4", 3), ("angle", "<f4"), ("octave", "<i4"), ("valid", "u1"), ("claims", "u1"), ("pad", "u1", 2)])


def tracking_scene(seed=4000, n=2000, w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448,
                   motion=(0.0, 0.0, 0.8), dense=False):
    """Two consecutive frames of a synthetic client: the last frame's features carry map points (world positions, representative
    descriptors), the current frame sees the same scene from a moved camera with noisy keypoints / descriptors, some features
    already claimed.  Returns the flat views the projection matchers take."""
    rng = np.random.default_rng(seed)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    Tlw = np.eye(4, dtype=np.float64); Tlw[:3, :3] = _rot(0.01, 0.02, -0.01); Tlw[:3, 3] = [0.2, -0.1, 0.3]
    d = np.eye(4); d[:3, :3] = _rot(0.004, -0.006, 0.002); d[:3, 3] = -np.asarray(motion)        # camera moves by `motion`
    Tcw = d @ Tlw
    # last-frame keypoints and their map points
    span = 0.25 if dense else 1.0                         # dense: many features share search windows (long claim chains)
    uL = rng.uniform(30, 30 + (w - 60) * span, n); vL = rng.uniform(20, 20 + (h - 40) * span, n); z = rng.uniform(5, 45, n)
    Xc = np.stack([(uL - cx) * z / fx, (vL - cy) * z / fy, z], 1)
    Xw = (Tlw[:3, :3].T @ (Xc - Tlw[:3, 3]).T).T
    octv = rng.integers(0, 8, n)
    last = np.zeros(n, LAST_DTYPE)
    last["world"] = Xw.astype(np.float32); last["angle"] = rng.uniform(0, 360, n); last["octave"] = octv
    last["valid"] = (rng.random(n) < 0.8); last["claims"] = (rng.random(n) < 0.9)
    last_desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    # current frame: re-observations (noisy) of ~85 % of the points + unrelated features
    Xc2 = (Tcw[:3, :3] @ Xw.T).T + Tcw[:3, 3]
    u2 = fx * Xc2[:, 0] / Xc2[:, 2] + cx; v2 = fy * Xc2[:, 1] / Xc2[:, 2] + cy
    seen = (rng.random(n) < 0.85) & (u2 > 5) & (u2 < w - 5) & (v2 > 5) & (v2 < h - 5)
    m = int(seen.sum()); extra = n - m
    keys = np.zeros(n, KP_DTYPE)
    keys["x"][:m] = u2[seen] + rng.normal(0, 1.5, m); keys["y"][:m] = v2[seen] + rng.normal(0, 1.5, m)
    keys["octave"][:m] = np.clip(octv[seen] + rng.integers(-1, 2, m), 0, 7)
    keys["angle"][:m] = (last["angle"][seen] + rng.normal(0, 8, m)) % 360
    keys["x"][m:] = rng.uniform(0, w, extra); keys["y"][m:] = rng.uniform(0, h, extra); keys["octave"][m:] = rng.integers(0, 8, extra); keys["angle"][m:] = rng.uniform(0, 360, extra)
    bits = np.unpackbits(last_desc[seen], axis=1); bits ^= (rng.random(bits.shape) < 0.07).astype(np.uint8)
    desc = np.zeros((n, 32), np.uint8); desc[:m] = np.packbits(bits, axis=1); desc[m:] = rng.integers(0, 256, (extra, 32), dtype=np.uint8)
    ur = np.full(n, -1.0, np.float32)
    st = rng.random(n) < 0.7
    ur[:m] = np.where(st[:m], keys["x"][:m] - bf / Xc2[seen, 2] + rng.normal(0, 1.0, m), -1.0)
    perm = rng.permutation(n)                             # shuffle feature order
    keys, desc, ur = keys[perm], desc[perm], ur[perm]
    cur = dict(keys_un=keys, u_right=ur, desc=desc, claimed=(rng.random(n) < 0.05).astype(np.uint8),
               min_x=0.0, min_y=0.0, max_x=float(w), max_y=float(h), scale=scale)
    # map-point view for SearchByProjection(Frame, MapPoints): Frame::isInFrustum outputs of the same points
    mps = np.zeros(n, TRACKED_DTYPE)
    mps["proj_x"] = u2; mps["proj_y"] = v2; mps["proj_xr"] = u2 - bf / Xc2[:, 2]
    mps["view_cos"] = rng.choice([0.9, 0.9985], n).astype(np.float32); mps["level"] = octv
    mps["valid"] = (rng.random(n) < 0.75) & (u2 > 0) & (u2 < w) & (v2 > 0) & (v2 < h); mps["claims"] = (rng.random(n) < 0.95)
    return dict(cur=cur, Tcw=Tcw.astype(np.float32), Tlw=Tlw.astype(np.float32), last=last, last_desc=last_desc, mps=mps,
                fx=float(np.float32(fx)), fy=float(np.float32(fy)), cx=float(np.float32(cx)), cy=float(np.float32(cy)),
                bf=float(np.float32(bf)), mb=float(np.float32(bf) / np.float32(fx)))


MP_DTYPE = np.dtype([("world", "<f4", 3), ("normal", "<f4", 3), ("min_distance", "<f4"), ("max_distance", "<f4"), ("angle", "<f4"),
                     ("valid", "u1"), ("pad", "u1", 3)])


def keyframe_scene(seed=5000, n=2000, w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, baseline=(0.6, 0.05, 0.4),
                   valid_frac=0.85, span=1.0):
    """Two keyframes that observe a common set of map points (loop-closure / relocalisation / fusion candidates).
    Every feature i of KF1 holds map point i (MP_DTYPE view with distance invariance limits and normal as MapPoint keeps them);
    KF2 re-observes most of them with noise plus unrelated features and holds its own (noisy) map points.  Returns the flat
    views the keyframe-target matchers take, and a similarity (s12, R12, t12) close to the true relative pose."""
    rng = np.random.default_rng(seed)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    T1w = np.eye(4); T1w[:3, :3] = _rot(0.01, -0.015, 0.005); T1w[:3, 3] = [0.1, 0.05, -0.2]
    d = np.eye(4); d[:3, :3] = _rot(0.01, 0.02, -0.008); d[:3, 3] = -np.asarray(baseline)
    T2w = d @ T1w
    u1 = rng.uniform(30, 30 + (w - 60) * span, n); v1 = rng.uniform(20, 20 + (h - 40) * span, n); z = rng.uniform(6, 40, n)
    Xc1 = np.stack([(u1 - cx) * z / fx, (v1 - cy) * z / fy, z], 1)
    Xw = (T1w[:3, :3].T @ (Xc1 - T1w[:3, 3]).T).T
    O1 = -T1w[:3, :3].T @ T1w[:3, 3]; O2 = -T2w[:3, :3].T @ T2w[:3, 3]
    oct1 = rng.integers(0, 7, n)
    def kf_dict(keys, ur, desc):
        return dict(keys_un=keys, u_right=ur.astype(np.float32), desc=desc, min_x=0.0, min_y=0.0, max_x=float(w), max_y=float(h), scale=scale,
                    inv_level_sigma2=(1.0 / (scale * scale)).astype(np.float32), log_scale_factor=float(np.float32(np.log(np.float32(1.2)))),
                    fx=float(np.float32(fx)), fy=float(np.float32(fy)), cx=float(np.float32(cx)), cy=float(np.float32(cy)), bf=float(np.float32(bf)))
    def mp_view(Xw_, O, octv, angles, valid):
        m = np.zeros(len(Xw_), MP_DTYPE)
        PO = Xw_ - O; dist = np.linalg.norm(PO, axis=1)
        m["world"] = Xw_.astype(np.float32); m["normal"] = (PO / dist[:, None]).astype(np.float32)
        m["max_distance"] = (dist * scale[octv]).astype(np.float32); m["min_distance"] = (m["max_distance"] / scale[7]).astype(np.float32)
        m["angle"] = angles; m["valid"] = valid
        return m
    desc1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    keys1 = np.zeros(n, KP_DTYPE); keys1["x"] = u1; keys1["y"] = v1; keys1["octave"] = oct1; keys1["angle"] = rng.uniform(0, 360, n)
    ur1 = np.where(rng.random(n) < 0.7, u1 - bf / z, -1.0)
    kf1 = kf_dict(keys1, ur1, desc1)
    pts1 = mp_view(Xw, O1, oct1, keys1["angle"], (rng.random(n) < valid_frac).astype(np.uint8))
    # KF2: noisy re-observations of ~80 % of the points + unrelated features, shuffled
    Xc2 = (T2w[:3, :3] @ Xw.T).T + T2w[:3, 3]
    u2 = fx * Xc2[:, 0] / Xc2[:, 2] + cx; v2 = fy * Xc2[:, 1] / Xc2[:, 2] + cy
    seen = (rng.random(n) < 0.8) & (u2 > 5) & (u2 < w - 5) & (v2 > 5) & (v2 < h - 5)
    m = int(seen.sum()); extra = n - m
    keys2 = np.zeros(n, KP_DTYPE)
    keys2["x"][:m] = u2[seen] + rng.normal(0, 1.2, m); keys2["y"][:m] = v2[seen] + rng.normal(0, 1.2, m)
    keys2["octave"][:m] = np.clip(oct1[seen] + rng.integers(-1, 2, m), 0, 7); keys2["angle"][:m] = (keys1["angle"][seen] + rng.normal(0, 8, m)) % 360
    keys2["x"][m:] = rng.uniform(0, w, extra); keys2["y"][m:] = rng.uniform(0, h, extra); keys2["octave"][m:] = rng.integers(0, 8, extra); keys2["angle"][m:] = rng.uniform(0, 360, extra)
    bits = np.unpackbits(desc1[seen], axis=1); bits ^= (rng.random(bits.shape) < 0.05).astype(np.uint8)
    desc2 = np.zeros((n, 32), np.uint8); desc2[:m] = np.packbits(bits, axis=1); desc2[m:] = rng.integers(0, 256, (extra, 32), dtype=np.uint8)
    ur2 = np.full(n, -1.0); ur2[:m] = np.where(rng.random(m) < 0.7, keys2["x"][:m] - bf / Xc2[seen, 2] + rng.normal(0, 0.8, m), -1.0)
    X2 = np.zeros((n, 3)); X2[:m] = Xw[seen] + rng.normal(0, 0.02, (m, 3))
    zz = rng.uniform(6, 40, extra); Xe = np.stack([(keys2["x"][m:] - cx) * zz / fx, (keys2["y"][m:] - cy) * zz / fy, zz], 1)
    X2[m:] = (T2w[:3, :3].T @ (Xe - T2w[:3, 3]).T).T
    src = np.full(n, -1); src[:m] = np.nonzero(seen)[0]
    perm = rng.permutation(n)
    keys2, desc2, ur2, X2, src = keys2[perm], desc2[perm], ur2[perm], X2[perm], src[perm]
    kf2 = kf_dict(keys2, ur2, desc2)
    pts2 = mp_view(X2, O2, keys2["octave"].astype(np.int64).clip(0, 6), keys2["angle"], (rng.random(n) < valid_frac).astype(np.uint8))
    bits2 = np.unpackbits(desc2, axis=1); bits2 ^= (rng.random(bits2.shape) < 0.02).astype(np.uint8)
    mp_desc2 = np.packbits(bits2, axis=1)                                   # representative descriptors of KF2's map points
    # similarity 1 <- 2 (p_c1 = s12 R12 p_c2 + t12), perturbed
    T12 = T1w @ np.linalg.inv(T2w)
    R12 = T12[:3, :3] @ _rot(0.002, -0.001, 0.0015); t12 = T12[:3, 3] + rng.normal(0, 0.01, 3); s12 = 1.0 + rng.normal(0, 0.01)
    return dict(kf1=kf1, kf2=kf2, T1w=T1w.astype(np.float32), T2w=T2w.astype(np.float32), Ow2=O2.astype(np.float32),
                pts1=pts1, desc1=desc1, pts2=pts2, desc2=mp_desc2, s12=float(np.float32(s12)), R12=R12.astype(np.float32), t12=t12.astype(np.float32),
                claimed2=(rng.random(n) < 0.1).astype(np.uint8), true_src2=src)


def crowd_keyframe_scene(sc, seed=0, group=3, frac=0.5):
    """Repeated texture for the ORDER-DEPENDENT matchers (SearchByProjection(KeyFrame*, Scw, ...), the relocalisation projection): groups of `group` map points of
    keyframe_scene()'s KF1 are moved next to their leader and given its descriptor (2 % of the bits flipped), and the features of KF2 that observe them are moved
    next to the leader's feature with the same descriptor (4 % flipped) -- every point of a group then finds several features within TH_LOW and the feature a point
    takes depends on what the earlier points of the call took.  Returns a modified copy."""
    rng = np.random.default_rng(seed)
    out = dict(sc)
    pts = sc["pts1"].copy(); desc1 = sc["desc1"].copy()
    kf2 = dict(sc["kf2"]); keys2 = kf2["keys_un"].copy(); desc2 = kf2["desc"].copy()
    src = sc["true_src2"]; n = len(pts)
    feat_of = np.full(n, -1); feat_of[src[src >= 0]] = np.nonzero(src >= 0)[0]
    cand = rng.permutation(np.nonzero(feat_of >= 0)[0])
    ng = int(len(cand) * frac) // group

    def flip(d, p):
        b = np.unpackbits(d); b ^= (rng.random(b.shape) < p).astype(np.uint8); return np.packbits(b)
    for g in range(ng):
        a = cand[g * group]; fa = feat_of[a]
        for b in cand[g * group + 1: (g + 1) * group]:
            fb = feat_of[b]
            pts["world"][b] = pts["world"][a] + rng.normal(0, 0.01, 3).astype(np.float32)
            for k in ("normal", "min_distance", "max_distance"):
                pts[k][b] = pts[k][a]
            desc1[b] = flip(desc1[a], 0.02)
            keys2["x"][fb] = keys2["x"][fa] + np.float32(rng.uniform(-3, 3)); keys2["y"][fb] = keys2["y"][fa] + np.float32(rng.uniform(-3, 3))
            keys2["octave"][fb] = keys2["octave"][fa]
            desc2[fb] = flip(desc1[a], 0.04)
        desc2[fa] = flip(desc1[a], 0.04)
    kf2["keys_un"] = keys2; kf2["desc"] = desc2
    out["pts1"] = pts; out["desc1"] = desc1; out["kf2"] = kf2
    return out


def monocular_init_pair(seed=7000, n=1500, span=1.0, crowd=False, steal_frac=0.08):
    """Two frames for ORBmatcher::SearchForInitialization (ORBmatcher.cc:540-655): keyframe_scene()'s KF1 / KF2 with most features at level 0 (the matcher reads level 0
    only) and vbPrevMatched = F1's own keypoints (Tracking.cc:585-587).  steal_frac of F1's matched level-0 features get a LATER twin -- a feature further down F1's list, at
    the same place, whose descriptor is closer to the F2 feature than theirs -- so that the sequential loop takes matches away from earlier features (:583, :601-605).
    Returns (f1, f2, prev_matched, true_src2)."""
    rng = np.random.default_rng(seed + 17)
    sc = keyframe_scene(seed, n=n, span=span)
    if crowd:
        sc = crowd_keyframe_scene(sc, seed)
    f1 = dict(sc["kf1"]); f2 = dict(sc["kf2"])
    k1 = f1["keys_un"].copy(); d1 = f1["desc"].copy(); k1["octave"] = np.where(np.arange(n) % 3 == 0, 1, 0)
    src = sc["true_src2"]
    k2 = f2["keys_un"].copy(); k2["octave"] = np.where(src >= 0, k1["octave"][np.maximum(src, 0)], np.arange(n) % 2)
    feat_of = np.full(n, -1); feat_of[src[src >= 0]] = np.nonzero(src >= 0)[0]
    early = np.nonzero((feat_of >= 0) & (k1["octave"] == 0) & (np.arange(n) < n // 2))[0]
    late = np.nonzero((k1["octave"] == 0) & (np.arange(n) >= n // 2))[0]
    m = min(int(len(early) * steal_frac), len(late))
    for a, b in zip(rng.choice(early, m, replace=False), rng.choice(late, m, replace=False)):
        bits = np.unpackbits(f2["desc"][feat_of[a]]); bits ^= (rng.random(256) < 0.01).astype(np.uint8)
        d1[b] = np.packbits(bits); k1["x"][b] = k1["x"][a] + np.float32(0.5); k1["y"][b] = k1["y"][a] - np.float32(0.5); k1["angle"][b] = k1["angle"][a]
    f1["keys_un"] = k1; f1["desc"] = d1; f2["keys_un"] = k2
    return f1, f2, np.stack([k1["x"], k1["y"]], 1).astype(np.float32), src


def sim3_problem(seed=6000, n=150, outlier_frac=0.12, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, w=1241, h=376, pix_noise=0.8,
                 scale=1.07, init_noise=(0.01, 0.05, 0.02)):
    """Loop-closure candidate for Optimizer::OptimizeSim3: n matched map points seen by two keyframes whose maps differ by a
    similarity (x1 = s12 R12 x2 + t12), a fraction of wrong matches, and a perturbed initial estimate (as Sim3Solver returns)."""
    rng = np.random.default_rng(seed)
    R12 = _rot(0.03, -0.05, 0.02); t12 = np.array([0.4, -0.1, 0.25]); s12 = scale
    octv1 = rng.integers(0, 6, n); octv2 = rng.integers(0, 6, n)
    u2 = rng.uniform(60, w - 60, n); v2 = rng.uniform(40, h - 40, n); z2 = rng.uniform(5, 35, n)
    X2 = np.stack([(u2 - cx) * z2 / fx, (v2 - cy) * z2 / fy, z2], 1)
    X1 = s12 * (R12 @ X2.T).T + t12
    X1n = X1 + rng.normal(0, 0.01, X1.shape)                                  # the two maps triangulated the point independently
    obs1 = np.stack([fx * X1[:, 0] / X1[:, 2] + cx, fy * X1[:, 1] / X1[:, 2] + cy], 1) + rng.normal(0, pix_noise, (n, 2))
    obs2 = np.stack([u2, v2], 1) + rng.normal(0, pix_noise, (n, 2))
    bad = rng.random(n) < outlier_frac
    obs1[bad] += rng.choice([-1, 1], (int(bad.sum()), 2)) * rng.uniform(15, 60, (int(bad.sum()), 2))
    sig = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    dR = _rot(*(rng.normal(0, init_noise[0], 3)))
    return dict(p1c=X1n.astype(np.float32), p2c=X2.astype(np.float32), obs1=obs1.astype(np.float32), obs2=obs2.astype(np.float32),
                inv_sigma2_1=(1.0 / sig[octv1] ** 2).astype(np.float32), inv_sigma2_2=(1.0 / sig[octv2] ** 2).astype(np.float32),
                fx1=fx, fy1=fy, cx1=cx, cy1=cy, fx2=fx, fy2=fy, cx2=cx, cy2=cy,
                R12=(dR @ R12).astype(np.float64), t12=(t12 + rng.normal(0, init_noise[1], 3)).astype(np.float64), s12=float(s12 * (1 + rng.normal(0, init_noise[2]))),
                R_true=R12, t_true=t12, s_true=s12, bad=bad)


# ---- Sim3 helpers (g2o::Sim3 layout: quaternion x y z w, translation, scale) for the essential-graph generator ----
def _q_from_R(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()
    return q if q[3] >= 0 else -q
def _q_rot(q, v):
    from scipy.spatial.transform import Rotation
    return Rotation.from_quat(q).apply(v)
def _q_mul(a, b):
    x1, y1, z1, w1 = a; x2, y2, z2, w2 = b
    return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2, w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
def sim3_mul(a, b):
    return np.concatenate([_q_mul(a[:4], b[:4]), a[7] * _q_rot(a[:4], b[4:7]) + a[4:7], [a[7] * b[7]]])
def sim3_inv(a):
    qc = np.array([-a[0], -a[1], -a[2], a[3]])
    return np.concatenate([qc, _q_rot(qc, (-1.0 / a[7]) * a[4:7]), [1.0 / a[7]]])
def sim3_from_T(T, s=1.0):
    return np.concatenate([_q_from_R(T[:3, :3]), T[:3, 3], [s]])


def essential_graph(seed=7000, K=60, radius=12.0, drift=(0.0015, 0.02, 0.004), covis=(2, 3), n_points=400):
    """Loop closure on a circular trajectory of K keyframes: odometry estimates with accumulated rotation / translation / scale
    drift, a Sim3 correction of the current keyframe and its 4 neighbours (CorrectedSim3), spanning-tree, covisibility and loop
    edges with measurements Sji = Sjw * Swi built as Optimizer::OptimizeEssentialGraph builds them.  Vertex 0 (the loop keyframe)
    is fixed.  Also map points attached to reference keyframes."""
    rng = np.random.default_rng(seed)
    Ttrue = []
    for k in range(K):
        a = 2 * np.pi * k / K * 0.97                                  # the loop almost closes
        Rwc = _rot(0, a, 0); c = np.array([radius * np.sin(a), 0.0, radius * (1 - np.cos(a))])
        T = np.eye(4); T[:3, :3] = Rwc.T; T[:3, 3] = -Rwc.T @ c; Ttrue.append(T)
    # odometry: relative motions with noise, scale drifting
    Test = [Ttrue[0].copy()]; sc = 1.0
    for k in range(1, K):
        rel = Ttrue[k] @ np.linalg.inv(Ttrue[k - 1])
        sc *= 1.0 + drift[2] * rng.normal(1.0, 0.3)
        n = np.eye(4); n[:3, :3] = _rot(*rng.normal(0, drift[0], 3)); n[:3, 3] = rng.normal(0, drift[1], 3)
        rel = n @ rel; rel[:3, 3] *= sc
        Test.append(rel @ Test[k - 1])
    non_corr = [sim3_from_T(T) for T in Test]                         # Siw with s = 1 (Optimizer.cc:884-888)
    cur = K - 1
    # corrected Sim3 of the current keyframe: what ComputeSim3 finds (close to the truth, scale = accumulated drift)
    Tc = Ttrue[cur].copy(); S_cur = sim3_from_T(Tc, 1.0); S_cur[4:7] *= sc; S_cur[7] = sc             # [sR | s t] maps world -> drifted camera frame
    S = np.stack(non_corr)
    corrected = {}
    for i in range(cur - 4, cur + 1):
        Sic = sim3_from_T(Test[i] @ np.linalg.inv(Test[cur]))
        corrected[i] = sim3_mul(Sic, S_cur)
        S[i] = corrected[i]
    vS = S.copy()                                                     # vScw
    vi, vj, meas = [], [], []
    def add(i, j, Sjw, Swi): vi.append(i); vj.append(j); meas.append(sim3_mul(Sjw, Swi))
    for i in corrected:                                               # loop edges (LoopConnections): corrected side i -> old keyframes
        for j in (0, 1, 2):
            add(i, j, vS[j], sim3_inv(vS[i]))
    for k in range(1, K):
        Swi = sim3_inv(non_corr[k])
        add(k, k - 1, non_corr[k - 1], Swi)                           # spanning tree
        for dk in covis:
            if k - dk >= 0 and not (k in corrected and k - dk in (0, 1, 2)):
                add(k, k - dk, non_corr[k - dk], Swi)                 # covisibility
    fixed = np.zeros(K, np.uint8); fixed[0] = 1
    ref = rng.integers(0, K, n_points).astype(np.int32); ref[::17] = -1
    pts = rng.normal(0, 5, (n_points, 3)).astype(np.float32)
    return dict(K=K, S=vS, fixed=fixed, vi=np.array(vi, np.int32), vj=np.array(vj, np.int32), meas=np.stack(meas),
                Ttrue=np.stack(Ttrue), Test=np.stack(Test), ref=ref, points=pts, scale_drift=sc)


KITTI_CAMS = {  # (fx, fy, cx, cy, bf) of the reference's stereo configurations
    "00-02": (718.856, 718.856, 607.1928, 185.2157, 386.1448),      # corbslam_client/Examples/Stereo/KITTI00-02.yaml:8-22
    "04-12": (707.0912, 707.0912, 601.8873, 183.1104, 379.8145),    # corbslam_client/Examples/Stereo/KITTI04-12.yaml:8-22
}


def ba_problem_fast(n_clients=8, kf_per_client=150, pts_per_kf=40, seed=1000, cams=None, w=1241, h=376, mono_frac=0.15, window=6,
                    shared_frac=0.02, pose_noise=(0.02, 0.0035), point_noise=0.05, pix_noise=1.0, obs_range=(3, 8), chunk=400000):
    """Vectorised generator of the fused multi-client global-BA problem of SURVEY s8d(ii) (same geometry as ba_problem, numpy instead of
    Python loops, so that 50 000 keyframes / 5 M points are generated in about a minute): client c drives a closed loop with 1 m spacing
    and carries camera cams[c % len(cams)]; every point is observed by the obs_range[0]..obs_range[1] nearest-in-time keyframes of its
    client that see it (mean ~5.5) and a `shared_frac` of the points also by 3 keyframes of the next client; 85 % stereo / 15 % mono
    edges; octaves drawn from the extractor's quota distribution; exactly one fixed pose (index 0 = mnId 1).  Returns the C-ABI arrays
    plus `intr` (n_poses x 5, the per-keyframe intrinsics)."""
    rng = np.random.default_rng(seed)
    cams = [KITTI_CAMS["00-02"]] if cams is None else list(cams)
    K = n_clients * kf_per_client
    R0 = max(kf_per_client / (2 * np.pi), 2.0)
    idx = np.arange(K); c_of = idx // kf_per_client; i_of = idx % kf_per_client
    th = 2 * np.pi * i_of / kf_per_client
    C = np.stack([c_of * 1.5 * R0 + R0 * np.cos(th), 0.3 * np.sin(3 * th), R0 * np.sin(th)], 1)
    fwd = np.stack([-np.sin(th), np.zeros(K), np.cos(th)], 1); down = np.tile(np.array([0.0, 1.0, 0.0]), (K, 1))
    right = np.cross(down, fwd); right /= np.linalg.norm(right, axis=1, keepdims=True)
    Rcw = np.stack([right, down, fwd], 1)                               # rows = camera axes in world coordinates
    tcw = -np.einsum("kij,kj->ki", Rcw, C)
    Tcw_true = np.zeros((K, 4, 4)); Tcw_true[:, :3, :3] = Rcw; Tcw_true[:, :3, 3] = tcw; Tcw_true[:, 3, 3] = 1
    intr = np.asarray([cams[c % len(cams)] for c in c_of], np.float32)
    intr64 = intr.astype(np.float64)
    sigma_oct = 1.2 ** np.arange(8)
    quota = np.array([434, 362, 302, 251, 209, 175, 145, 122], float); quota /= quota.sum()
    M = K * pts_per_kf
    pts = np.zeros((M, 3)); e_parts = []
    offs = np.concatenate([[0], np.repeat(np.arange(1, window + 1), 2) * np.tile([1, -1], window)])       # nearest in time first: 0, +1, -1, +2, -2, ...
    for m0 in range(0, M, chunk):
        m1 = min(M, m0 + chunk); n = m1 - m0
        kf = np.arange(m0, m1) // pts_per_kf; c = kf // kf_per_client
        z = rng.uniform(4.0, 30.0, n); u = rng.uniform(40, w - 40, n); v = rng.uniform(30, h - 30, n)
        cam = intr64[kf]
        Xc = np.stack([(u - cam[:, 2]) * z / cam[:, 0], (v - cam[:, 3]) * z / cam[:, 1], z], 1)
        Xw = np.einsum("nji,nj->ni", Rcw[kf], Xc - tcw[kf])
        pts[m0:m1] = Xw
        cand = c[:, None] * kf_per_client + (kf[:, None] - c[:, None] * kf_per_client + offs[None, :]) % kf_per_client      # (n, 2 window + 1)
        want = rng.integers(obs_range[0], obs_range[1] + 1, n)
        shared = (rng.random(n) < shared_frac) if n_clients > 1 else np.zeros(n, bool)
        extra = ((c[:, None] + 1) % n_clients) * kf_per_client + rng.integers(0, kf_per_client, (n, 3))
        cand = np.concatenate([cand, extra], 1); ncand = cand.shape[1]
        Xj = np.einsum("ncij,nj->nci", Rcw[cand], Xw) + tcw[cand]
        camj = intr64[cand]
        uu = camj[..., 0] * Xj[..., 0] / Xj[..., 2] + camj[..., 2]; vv = camj[..., 1] * Xj[..., 1] / Xj[..., 2] + camj[..., 3]
        vis = (Xj[..., 2] >= 1.0) & (uu >= 0) & (uu < w) & (vv >= 0) & (vv < h)
        vis[:, 0] = True                                                # the anchor keyframe always observes its point
        own = vis[:, : ncand - 3] & (np.cumsum(vis[:, : ncand - 3], 1) <= want[:, None])      # the `want` nearest-in-time visible keyframes
        sel = np.concatenate([own, vis[:, ncand - 3:] & shared[:, None]], 1)
        # a point needs two observations: add the next keyframe in time if only the anchor saw it
        lonely = sel.sum(1) < 2
        sel[lonely, 1] = Xj[lonely, 1, 2] > 0.5
        pi, ci = np.nonzero(sel)
        j = cand[pi, ci]; ne = len(pi)
        octv = rng.choice(8, size=ne, p=quota); s = sigma_oct[octv] * pix_noise
        un = uu[pi, ci] + rng.normal(0, 1, ne) * s; vn = vv[pi, ci] + rng.normal(0, 1, ne) * s
        ur = un - camj[pi, ci, 4] / Xj[pi, ci, 2] + rng.normal(0, 1, ne) * s * 0.5
        ur[rng.random(ne) < mono_frac] = -1.0
        ur = np.where((ur < 0) & (ur > -1.0), 0.0, ur)                  # (a stereo right coordinate that noise pushed below 0 would read as "mono")
        e = np.zeros(ne, EDGE_DTYPE)
        e["pose"] = j; e["point"] = m0 + pi; e["u"] = un; e["v"] = vn; e["ur"] = ur; e["inv_sigma2"] = 1.0 / sigma_oct[octv] ** 2
        e_parts.append(e)
    edges = np.concatenate(e_parts)
    # duplicate (pose, point) pairs (a shared point whose random foreign keyframe repeats): keep the first -- a keyframe observes a map point once
    key = edges["pose"].astype(np.int64) * M + edges["point"]
    _, first = np.unique(key, return_index=True)
    edges = edges[np.sort(first)]
    poses0 = Tcw_true.copy()
    ang = rng.normal(0, pose_noise[1], (K, 3)); dt = rng.normal(0, pose_noise[0], (K, 3))
    ang[0] = 0; dt[0] = 0
    cx_, sx_ = np.cos(ang[:, 0]), np.sin(ang[:, 0]); cy_, sy_ = np.cos(ang[:, 1]), np.sin(ang[:, 1]); cz_, sz_ = np.cos(ang[:, 2]), np.sin(ang[:, 2])
    one = np.ones(K); zero = np.zeros(K)
    Rx = np.stack([one, zero, zero, zero, cx_, -sx_, zero, sx_, cx_], 1).reshape(K, 3, 3)
    Ry = np.stack([cy_, zero, sy_, zero, one, zero, -sy_, zero, cy_], 1).reshape(K, 3, 3)
    Rz = np.stack([cz_, -sz_, zero, sz_, cz_, zero, zero, zero, one], 1).reshape(K, 3, 3)
    dR = Rz @ Ry @ Rx
    poses0[:, :3, :3] = dR @ Rcw; poses0[:, :3, 3] = np.einsum("kij,kj->ki", dR, tcw) + dt
    points0 = (pts + rng.normal(0, point_noise, pts.shape)).astype(np.float32)
    pose_fixed = np.zeros(K, np.uint8); pose_fixed[0] = 1
    cam0 = intr[0]
    return dict(poses=poses0.astype(np.float32), pose_fixed=pose_fixed, points=points0, point_fixed=np.zeros(M, np.uint8), edges=edges,
                fx=float(cam0[0]), fy=float(cam0[1]), cx=float(cam0[2]), cy=float(cam0[3]), bf=float(cam0[4]), intr=intr,
                poses_true=Tcw_true, points_true=pts)


def client_maps(prob, n_clients, kf_per_client, frames=None):
    """A fused multi-client BA problem (ba_problem / ba_problem_fast) as the per-client maps a CORB-SLAM server receives: for every client its keyframes
    (ids (c)*1000000 + i + 1 like KeyFrame.cc:49, so the first keyframe of client 0 is mnId 1; one feature per observation: keypoint = (u, v), octave
    from the edge's information, mvuRight; pose, per-keyframe intrinsics, mvInvLevelSigma2) and its map points (id, world position, mObservations =
    (keyframe id, feature index) ascending in keyframe id -- std::map order).  frames[c] = To2n of client c (4x4): the client's map is expressed in its
    own frame, i.e. MapFusion's re-basing with To2n brings it back (Tcw_client = Tcw * To2n^-1, p_client = Rcw p + tcw).
    Returns a list of dict(kf=[dict(id, kp, ur, mp_id, meta fields...)], mp_records, obs_off, obs_kf, obs_idx)."""
    from . import KP_DTYPE, MP_RECORD_DTYPE, NO_MAP_POINT
    e = prob["edges"]; K = len(prob["poses"]); M = len(prob["points"])
    pts_per_kf = M // K
    inv_s2 = (1.0 / (1.2 ** np.arange(8)) ** 2).astype(np.float32)
    kf_id = lambda k: (k // kf_per_client) * 1000000 + (k % kf_per_client) + 1
    mp_id = lambda m: ((m // pts_per_kf) // kf_per_client) * 1000000 + (m % (pts_per_kf * kf_per_client)) + 1
    order = np.argsort(e["pose"], kind="stable")
    feat = np.zeros(len(e), np.int64)                       # feature index of every observation inside its keyframe
    starts = np.searchsorted(e["pose"][order], np.arange(K + 1))
    for k in range(K):
        feat[order[starts[k]: starts[k + 1]]] = np.arange(starts[k + 1] - starts[k])
    intr = prob.get("intr")
    out = []
    for c in range(n_clients):
        inv = np.linalg.inv(frames[c].astype(np.float64)) if frames is not None else np.eye(4)
        fr = frames[c].astype(np.float64) if frames is not None else np.eye(4)
        kfs = []
        for k in range(c * kf_per_client, (c + 1) * kf_per_client):
            ee = e[order[starts[k]: starts[k + 1]]]
            kp = np.zeros(len(ee), KP_DTYPE); kp["x"] = ee["u"]; kp["y"] = ee["v"]; kp["size"] = 31.0; kp["angle"] = 0.0
            kp["octave"] = np.argmin(np.abs(inv_s2[None, :] - ee["inv_sigma2"][:, None]), 1)
            cam = intr[k] if intr is not None else np.array([prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"]], np.float32)
            T = (prob["poses"][k].reshape(4, 4).astype(np.float64) @ inv).astype(np.float32)
            kfs.append(dict(id=kf_id(k), client_id=c + 1, kp=kp, ur=ee["ur"].astype(np.float32), mp_id=np.array([mp_id(m) for m in ee["point"]], np.uint64),
                            Tcw=T, cam=cam, inv_level_sigma2=inv_s2, desc=np.zeros((len(ee), 32), np.uint8)))
        m0, m1 = c * kf_per_client * pts_per_kf, (c + 1) * kf_per_client * pts_per_kf
        rec = np.zeros(m1 - m0, MP_RECORD_DTYPE)
        sel = np.flatnonzero((e["point"] >= m0) & (e["point"] < m1))
        ids = np.array([kf_id(k) for k in e["pose"][sel]], np.uint64)
        o2 = np.lexsort((ids, e["point"][sel]))              # by map point, then ascending keyframe id
        sel = sel[o2]; ids = ids[o2]
        cnt = np.bincount(e["point"][sel] - m0, minlength=m1 - m0)
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        rec["id"] = [mp_id(m) for m in range(m0, m1)]; rec["client_id"] = c + 1; rec["n_obs"] = cnt
        rec["ref_kf_id"] = [kf_id(m // pts_per_kf) for m in range(m0, m1)]
        p = prob["points"][m0:m1].astype(np.float64)
        rec["world_pos"] = (p @ fr[:3, :3].T + fr[:3, 3]).astype(np.float32)
        rec["flags"] = np.where(prob["point_fixed"][m0:m1] != 0, 2, 0)
        rec["min_distance"] = 1.0; rec["max_distance"] = 50.0
        out.append(dict(kf=kfs, mp_records=rec, obs_off=off, obs_kf=ids, obs_idx=feat[sel].astype(np.uint32)))
    return out


def map_arrays(prob, kf_per_client, pts_per_kf):
    """The fused BA problem of ba_problem_fast as the flat arrays corb_kf_store_put_batch / corb_mp_store_put_host take (vectorised: 50 000 keyframes in seconds):
    keyframe k gets id (k // kf_per_client) * 1000000 + k % kf_per_client + 1, one feature per observation in edge order (keypoint = (u, v), octave from the edge's
    information, mvuRight); map point m gets id, world position and its observations (keyframe id, feature index) ascending in keyframe id (std::map order)."""
    from . import KP_DTYPE, MP_RECORD_DTYPE, KF_META_DTYPE
    e = prob["edges"]; K = len(prob["poses"]); M = len(prob["points"])
    inv_s2 = (1.0 / (1.2 ** np.arange(8)) ** 2).astype(np.float32)
    karr = np.arange(K, dtype=np.int64); kid = (karr // kf_per_client) * 1000000 + karr % kf_per_client + 1
    marr = np.arange(M, dtype=np.int64); per_client = kf_per_client * pts_per_kf
    mid = (marr // per_client) * 1000000 + marr % per_client + 1
    order = np.argsort(e["pose"], kind="stable")
    ek = e[order]
    feat_off = np.concatenate([[0], np.cumsum(np.bincount(ek["pose"], minlength=K))]).astype(np.int32)
    feat_of_edge = np.empty(len(e), np.int64); feat_of_edge[order] = np.arange(len(e)) - feat_off[ek["pose"]]
    kp = np.zeros(len(e), KP_DTYPE); kp["x"] = ek["u"]; kp["y"] = ek["v"]; kp["size"] = 31.0
    kp["octave"] = np.clip(np.rint(np.log(1.0 / ek["inv_sigma2"].astype(np.float64)) / np.log(1.44)), 0, 7).astype(np.int32)
    meta = np.zeros(K, KF_META_DTYPE)
    meta["id"] = kid; meta["client_id"] = karr // kf_per_client + 1; meta["nlevels"] = 8
    intr = prob.get("intr")
    cam = intr if intr is not None else np.tile(np.array([prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"]], np.float32), (K, 1))
    for a, name in enumerate(("fx", "fy", "cx", "cy", "bf")):
        meta[name] = cam[:, a]
    meta["Tcw"] = prob["poses"].reshape(K, 16); meta["TcwGBA"] = np.eye(4, dtype=np.float32).reshape(16)
    meta["inv_level_sigma2"][:, :8] = inv_s2
    meta["flags"] = np.where(prob["pose_fixed"] != 0, 2, 0); meta["flags"][0] = 0          # (keyframe 0 is fixed by its id, mnId == 1)
    rec = np.zeros(M, MP_RECORD_DTYPE)
    rec["id"] = mid; rec["client_id"] = marr // per_client + 1; rec["ref_kf_id"] = kid[marr // pts_per_kf]; rec["world_pos"] = prob["points"]
    rec["flags"] = np.where(prob["point_fixed"] != 0, 2, 0); rec["min_distance"] = 1.0; rec["max_distance"] = 50.0
    ids = kid[e["pose"]].astype(np.uint64)
    o2 = np.lexsort((ids, e["point"]))
    obs_off = np.concatenate([[0], np.cumsum(np.bincount(e["point"], minlength=M))]).astype(np.int32)
    rec["n_obs"] = np.diff(obs_off)
    return dict(meta=meta, feat_off=feat_off, kp=kp, ur=ek["ur"].astype(np.float32), mp_id=mid[ek["point"]].astype(np.uint64),
                mp_records=rec, obs_off=obs_off, obs_kf=ids[o2], obs_idx=feat_of_edge[o2].astype(np.uint32), max_features=int(np.diff(feat_off).max()), max_obs=int(rec["n_obs"].max()))


_TRACKED_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"), ("level", "<i4"), ("valid", "u1"), ("claims", "u1"), ("pad", "u1", 2)])


def frustum_view(Tcw, world, normal, min_distance, max_distance, fx, fy, cx, cy, bf, min_x, max_x, min_y, max_y, log_scale_factor, nlevels, cos_limit=0.5):
    """What the CALLER of corb_search_by_projection_map computes on the CPU in the reference -- bool Frame::isInFrustum(MapPoint*, viewingCosLimit) for an array of MapPoints (corbslam_client/src/Frame.cc:270-329; MapPoint::PredictScale
    corbslam_client/src/MapPoint.cc:500-514; mOw = -mRcw.t()*mtcw, Frame.cc UpdatePoseMatrices) in numpy with the reference's arithmetic: float operands,
    cv::gemm / cv::norm / Mat::dot accumulate in double and round once, libm log on a float DEFINED as (float)log((double)x) like orc_proj.c:217.
    Returns a _TRACKED_DTYPE array (valid = mbTrackInView; claims left 0)."""
    f = np.float32
    T = np.asarray(Tcw, f).reshape(4, 4); P = np.asarray(world, f).reshape(-1, 3); Pn = np.asarray(normal, f).reshape(-1, 3)
    R = T[:3, :3].astype(np.float64); t = T[:3, 3].astype(np.float64)
    Pd = P.astype(np.float64)
    Pc = np.stack([((R[i, 0] * Pd[:, 0] + R[i, 1] * Pd[:, 1]) + R[i, 2] * Pd[:, 2]) + t[i] for i in range(3)], 1).astype(f)
    Ow = np.array([-((R[0, i] * t[0] + R[1, i] * t[1]) + R[2, i] * t[2]) for i in range(3)]).astype(f)
    out = np.zeros(len(P), _TRACKED_DTYPE)
    with np.errstate(divide="ignore", invalid="ignore"):
        ok = Pc[:, 2] > f(0)
        invz = f(1) / Pc[:, 2]
        u = (f(fx) * Pc[:, 0]) * invz + f(cx); v = (f(fy) * Pc[:, 1]) * invz + f(cy)
        ok &= ~((u < f(min_x)) | (u > f(max_x)) | (v < f(min_y)) | (v > f(max_y)))
        maxD = f(1.2) * np.asarray(max_distance, f); minD = f(0.8) * np.asarray(min_distance, f)
        PO = P - Ow
        POd = PO.astype(np.float64)
        dist = np.sqrt((POd[:, 0] * POd[:, 0] + POd[:, 1] * POd[:, 1]) + POd[:, 2] * POd[:, 2]).astype(f)
        ok &= ~((dist < minD) | (dist > maxD))
        Pnd = Pn.astype(np.float64)
        dot = (POd[:, 0] * Pnd[:, 0] + POd[:, 1] * Pnd[:, 1]) + POd[:, 2] * Pnd[:, 2]
        view = (dot / dist.astype(np.float64)).astype(f)
        ok &= ~(view < f(cos_limit))
        ratio = np.asarray(max_distance, f) / dist
        lg = np.log(ratio.astype(np.float64)).astype(f)
        lvl = np.ceil(lg / f(log_scale_factor))
    lvl = np.where(np.isfinite(lvl), lvl, 0).astype(np.int64)
    lvl = np.clip(lvl, 0, nlevels - 1)
    out["valid"] = ok
    out["proj_x"] = np.where(ok, u, 0); out["proj_y"] = np.where(ok, v, 0); out["proj_xr"] = np.where(ok, u - f(bf) * invz, 0)
    out["view_cos"] = np.where(ok, view, 0); out["level"] = np.where(ok, lvl, 0)
    return out
