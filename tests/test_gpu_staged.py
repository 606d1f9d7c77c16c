"""GPU parity: LocalBundleAdjustment / PoseOptimization through corb_ba_solve_staged vs the oracle (1e-4 relative,
identical outlier classification)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [2000, 2001, 2002])
def test_local_bundle_adjustment(corb, pyorc, synth, seed):
    p = synth.local_ba_problem(seed=seed)
    args = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    g = corb.Optimizer.LocalBundleAdjustment(*args)
    r = pyorc.ba_solve_staged(*args, pyorc.LOCAL_BA_STAGES)
    assert g["iters_done"] == r["iters_done"] and g["trials"] == r["trials"]
    assert np.array_equal(g["outlier"], r["outlier"])
    assert np.abs(g["poses"] - r["poses"]).max() <= 1e-4 * max(1.0, np.abs(r["poses"]).max())
    assert np.abs(g["points"] - r["points"]).max() <= 1e-4 * max(1.0, np.abs(r["points"]).max())
    assert g["outlier"].sum() > 0


@pytest.mark.parametrize("seed,kw", [(2020, dict(pts_per_kf=140)), (2021, dict(n_local=10, n_fixed=8, pts_per_kf=90, outlier_frac=0.12)),
                                     (2022, dict(n_local=30, n_fixed=4, pts_per_kf=30))])        # the last: 29 free keyframes -- no one-workgroup solve, no LM chains
def test_local_window_on_the_device_equals_the_host_route_bit_for_bit(corb, pyorc, synth, seed, kw):
    """A window whose edges come grouped by point (the order Optimizer.cc creates them in) is flattened, optimised and classified on the device
    (corb_ba.cpp: ba_staged_window_host); the same window with ONE point's edges moved to the end is no longer grouped and takes the host flattening.  The stable sort
    of the host flattening puts every landmark's edges back in the same order, so both routes run the same sums: the same bits, the same flags -- and the oracle's."""
    if os.environ.get("CORB_LBA_HOST_FLATTEN") is not None:
        pytest.skip("the library was told to take the host route everywhere (development switch)")
    p = synth.local_ba_problem(seed=seed, **kw)
    e = p["edges"]
    assert np.all(np.diff(e["point"]) >= 0) and len(e) > 2048                         # grouped; beyond the one-workgroup optimiser
    moved = e["point"] == e["point"][0]
    order = np.r_[np.nonzero(~moved)[0], np.nonzero(moved)[0]]
    a = lambda edges: (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], edges, p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    g = corb.Optimizer.LocalBundleAdjustment(*a(e))
    h = corb.Optimizer.LocalBundleAdjustment(*a(e[order]))
    assert g["device_route"] and not h["device_route"]
    assert g["iters_done"] == h["iters_done"] and g["trials"] == h["trials"]
    assert np.array_equal(g["poses"], h["poses"]) and np.array_equal(g["points"], h["points"])
    assert np.array_equal(g["outlier"][order], h["outlier"]) and g["outlier"].sum() > 0
    r = pyorc.ba_solve_staged(*a(e), pyorc.LOCAL_BA_STAGES)
    assert g["iters_done"] == r["iters_done"] and g["trials"] == r["trials"] and np.array_equal(g["outlier"], r["outlier"])
    assert np.abs(g["poses"] - r["poses"]).max() <= 1e-4 * max(1.0, np.abs(r["poses"]).max()) and np.abs(g["points"] - r["points"]).max() <= 1e-4 * max(1.0, np.abs(r["points"]).max())


def test_local_window_routes_agree_on_fixed_points_and_vertices_without_edges(corb, synth):
    """the same comparison on a window with fixed map points, map points without an observation and a free keyframe without one: untouched vertices keep their input
    floats on both routes, fixed landmarks' edges sit behind the free ones'"""
    if os.environ.get("CORB_LBA_HOST_FLATTEN") is not None:
        pytest.skip("the library was told to take the host route everywhere (development switch)")
    p = synth.local_ba_problem(seed=2030, n_local=7, n_fixed=4, pts_per_kf=120, outlier_frac=0.08)
    rng = np.random.default_rng(5)
    e = p["edges"]
    point_fixed = p["point_fixed"].copy(); point_fixed[rng.random(len(point_fixed)) < 0.05] = 1
    drop_pts = rng.choice(len(p["points"]), 6, replace=False)
    keep = ~np.isin(e["point"], drop_pts) & (e["pose"] != 3)            # six points and the free keyframe 3 lose every observation
    e = e[keep]
    assert np.all(np.diff(e["point"]) >= 0) and len(e) > 2048 and not p["pose_fixed"][3]
    moved = e["point"] == e["point"][0]
    order = np.r_[np.nonzero(~moved)[0], np.nonzero(moved)[0]]
    a = lambda edges: (p["poses"], p["pose_fixed"], p["points"], point_fixed, edges, p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    g = corb.Optimizer.LocalBundleAdjustment(*a(e))
    h = corb.Optimizer.LocalBundleAdjustment(*a(e[order]))
    assert g["device_route"] and not h["device_route"]
    assert g["iters_done"] == h["iters_done"] and g["trials"] == h["trials"]
    assert np.array_equal(g["poses"], h["poses"]) and np.array_equal(g["points"], h["points"]) and np.array_equal(g["outlier"][order], h["outlier"])
    assert np.array_equal(g["poses"][3].reshape(16), p["poses"][3].reshape(16)) and np.array_equal(g["points"][drop_pts], p["points"][drop_pts])
    assert np.array_equal(g["points"][point_fixed != 0], p["points"][point_fixed != 0])
    #  an edge between a FIXED keyframe and a FIXED point is outside the graph g2o optimises, yet the classification's isDepthPositive() still sees it: put one such
    # point behind a fixed camera that observes it -- both routes flag exactly that observation (and agree on everything else)
    ff = np.nonzero((p["pose_fixed"][e["pose"]] != 0) & (point_fixed[e["point"]] != 0))[0]
    assert len(ff) > 0
    j, k = int(e["point"][ff[0]]), int(e["pose"][ff[0]])
    pts2 = p["points"].copy(); T = p["poses"][k].reshape(4, 4).astype(np.float64)
    pts2[j] = (T[:3, :3].T @ (np.array([0.3, -0.2, -4.0]) - T[:3, 3])).astype(np.float32)
    a2 = lambda edges: (p["poses"], p["pose_fixed"], pts2, point_fixed, edges, p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    g2 = corb.Optimizer.LocalBundleAdjustment(*a2(e)); h2 = corb.Optimizer.LocalBundleAdjustment(*a2(e[order]))
    assert g2["device_route"] and not h2["device_route"]
    assert g2["outlier"][ff[0]] == 1 and np.array_equal(g2["outlier"][order], h2["outlier"]) and np.array_equal(g2["poses"], h2["poses"]) and np.array_equal(g2["points"], h2["points"])


def test_local_window_routes_agree_on_random_windows(corb, synth):
    """ten random windows (keyframe counts, fixed keyframes incl. none, fixed points, thinned observations): the device route and the host route, bit for bit"""
    if os.environ.get("CORB_LBA_HOST_FLATTEN") is not None:
        pytest.skip("the library was told to take the host route everywhere (development switch)")
    rng = np.random.default_rng(77)
    done = 0
    for seed in range(2200, 2216):
        n_local = int(rng.integers(2, 13)); n_fixed = int(rng.integers(0, 7)); ppk = int(rng.integers(110, 170))
        p = synth.local_ba_problem(seed=seed, n_local=n_local, n_fixed=n_fixed, pts_per_kf=ppk, outlier_frac=float(rng.uniform(0.02, 0.15)))
        e = p["edges"]
        e = e[rng.random(len(e)) > rng.uniform(0.0, 0.2)]                      # thinned: some points lose observations, a few lose all of them
        point_fixed = p["point_fixed"].copy(); point_fixed[rng.random(len(point_fixed)) < rng.uniform(0.0, 0.1)] = 1
        if len(e) <= 2100 or not np.all(np.diff(e["point"]) >= 0) or (p["pose_fixed"] == 0).sum() < 1:
            continue
        moved = e["point"] == e["point"][len(e) // 2]
        order = np.r_[np.nonzero(~moved)[0], np.nonzero(moved)[0]]
        a = lambda edges: (p["poses"], p["pose_fixed"], p["points"], point_fixed, edges, p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
        g = corb.Optimizer.LocalBundleAdjustment(*a(e))
        h = corb.Optimizer.LocalBundleAdjustment(*a(e[order]))
        assert not h["device_route"], seed
        if not g["device_route"]:                                              # (a window at the one-workgroup optimiser's size is declined: both calls took the host route)
            continue
        assert g["iters_done"] == h["iters_done"] and g["trials"] == h["trials"], seed
        assert np.array_equal(g["poses"], h["poses"]) and np.array_equal(g["points"], h["points"]) and np.array_equal(g["outlier"][order], h["outlier"]), seed
        done += 1
    assert done >= 8


@pytest.mark.parametrize("seed", [3000, 3001, 3002, 3003])
def test_pose_optimization(corb, pyorc, synth, seed):
    q = synth.pose_opt_problem(seed=seed, n=300 + 50 * (seed % 4))
    T, outl, ninl = corb.Optimizer.PoseOptimization(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    n = len(q["points"])
    edges = np.zeros(n, pyorc.EDGE_DTYPE)
    edges["pose"] = 0; edges["point"] = np.arange(n); edges["u"] = q["obs"][:, 0]; edges["v"] = q["obs"][:, 1]; edges["ur"] = q["obs"][:, 2]
    edges["inv_sigma2"] = q["inv_sigma2"]
    r = pyorc.ba_solve_staged(q["Tcw0"].reshape(1, 16), np.zeros(1, np.uint8), q["points"], np.ones(n, np.uint8), edges,
                              q["fx"], q["fy"], q["cx"], q["cy"], q["bf"], pyorc.POSE_OPT_STAGES)
    assert np.array_equal(outl, r["outlier"].astype(bool)) and ninl == n - int(r["outlier"].sum())
    assert np.abs(T - r["poses"][0]).max() <= 1e-4 * max(1.0, np.abs(r["poses"][0]).max())
    assert np.abs(T[:3, 3] - q["Tcw_true"][:3, 3]).max() < 0.03


@pytest.mark.parametrize("n", [511, 513, 1200, 1750, 2048, 2049, 3000])
def test_pose_optimization_sizes(corb, pyorc, synth, n):
    """The kernel keeps up to 4 edges per thread (512 threads) in registers and re-reads them from memory above 2 048 per frame: sizes on both sides
    of every boundary (1 750 = a KITTI frame's matches; 3 000 = the 4 000-feature configuration), singly and as one batch."""
    q = synth.pose_opt_problem(seed=3040 + n, n=n)
    T, outl, ninl = corb.Optimizer.PoseOptimization(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    r = _oracle_pose_opt(pyorc, q)
    assert np.array_equal(outl, r["outlier"].astype(bool)) and ninl == n - int(r["outlier"].sum())
    assert np.abs(T - r["poses"][0]).max() <= 1e-4 * max(1.0, np.abs(r["poses"][0]).max())
    q2 = synth.pose_opt_problem(seed=3041, n=300)
    res = corb.Optimizer.PoseOptimizationBatch([(q2["Tcw0"], q2["points"], q2["obs"], q2["inv_sigma2"]), (q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"])],
                                               q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    assert np.array_equal(res[1][0], T) and np.array_equal(res[1][1], outl) and res[1][2] == ninl


def _oracle_pose_opt(pyorc, q):
    n = len(q["points"])
    edges = np.zeros(n, pyorc.EDGE_DTYPE)
    edges["pose"] = 0; edges["point"] = np.arange(n); edges["u"] = q["obs"][:, 0]; edges["v"] = q["obs"][:, 1]; edges["ur"] = q["obs"][:, 2]
    edges["inv_sigma2"] = q["inv_sigma2"]
    return pyorc.ba_solve_staged(q["Tcw0"].reshape(1, 16), np.zeros(1, np.uint8), q["points"], np.ones(n, np.uint8), edges,
                                 q["fx"], q["fy"], q["cx"], q["cy"], q["bf"], pyorc.POSE_OPT_STAGES)


@pytest.mark.parametrize("seed", [3000, 3003])
def test_pose_optimization_general_path_agrees(corb, pyorc, synth, seed):
    """solver=1 (general staged path: rocSOLVER 6x6, host LM control) and the fused kernel give the same classification."""
    q = synth.pose_opt_problem(seed=seed, n=300 + 50 * (seed % 4))
    a = (q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    T0, o0, n0 = corb.Optimizer.PoseOptimization(*a, solver=3)
    T1, o1, n1 = corb.Optimizer.PoseOptimization(*a, solver=1)
    assert np.array_equal(o0, o1) and n0 == n1
    assert np.abs(T0 - T1).max() <= 1e-4 * max(1.0, np.abs(T1).max())


def test_pose_optimization_batch(corb, pyorc, synth):
    """corb_pose_optimization_batch: one workgroup per frame; every frame equals its oracle run (incl. an empty frame and
    a frame whose observations are all gross outliers)."""
    qs = [synth.pose_opt_problem(seed=3100 + i, n=150 + 40 * i) for i in range(6)]
    frames = [(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"]) for q in qs]
    empty = (qs[0]["Tcw0"], np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.float32))
    frames.append(empty)
    res = corb.Optimizer.PoseOptimizationBatch(frames, qs[0]["fx"], qs[0]["fy"], qs[0]["cx"], qs[0]["cy"], qs[0]["bf"])
    assert len(res) == 7
    for q, (T, outl, ninl) in zip(qs, res[:6]):
        r = _oracle_pose_opt(pyorc, q)
        assert np.array_equal(outl, r["outlier"].astype(bool)) and ninl == len(q["points"]) - int(r["outlier"].sum())
        assert np.abs(T.reshape(16) - r["poses"][0].reshape(16)).max() <= 1e-4 * max(1.0, np.abs(r["poses"][0]).max())
    T, outl, ninl = res[6]
    assert ninl == 0 and len(outl) == 0 and np.array_equal(T.reshape(16), np.asarray(qs[0]["Tcw0"], np.float32).reshape(16))


def test_local_ba_stop_flag_semantics(corb, pyorc, synth):
    """pbStopFlag of LocalBundleAdjustment (LocalMapping::InterruptBA raises it routinely): before the first optimize() -> nothing changes; during the
    first round -> second round skipped, final test on every edge + write-back still run (Optimizer.cc:706-800).  Same outlier set / estimates as the oracle."""
    for seed, kw in ((2011, {}), (2012, dict(n_local=12, n_fixed=6, pts_per_kf=40))):       # fused one-workgroup optimiser / multi-kernel path
        p = synth.local_ba_problem(seed=seed, **kw)
        a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
        g0 = corb.Optimizer._staged(corb.LOCAL_BA_STAGES, *a, stop="before")
        assert np.array_equal(g0["poses"].reshape(-1, 16), p["poses"].reshape(-1, 16)) and np.array_equal(g0["points"], p["points"]) and g0["outlier"].sum() == 0
        g1 = corb.Optimizer._staged(corb.LOCAL_BA_STAGES, *a, stop="after_first_stage")
        r1 = pyorc.ba_solve_staged(*a, pyorc.LOCAL_BA_STAGES, stop="after_first_stage")
        full = corb.Optimizer._staged(corb.LOCAL_BA_STAGES, *a)
        assert g1["iters_done"] == r1["iters_done"] < full["iters_done"]
        assert np.array_equal(g1["outlier"], r1["outlier"]) and g1["outlier"].sum() > 0
        assert np.abs(g1["poses"] - r1["poses"]).max() < 1e-4 and np.abs(g1["points"] - r1["points"]).max() < 1e-3


def test_pose_optimization_early_outs(corb, pyorc, synth):
    """Optimizer::PoseOptimization: fewer than 3 correspondences -> plain `return 0`, pose and flags untouched (Optimizer.cc:396-397); fewer than 10
    edges in the graph -> only the first of the four rounds runs (:470-471).  In one batch with a regular frame."""
    q = synth.pose_opt_problem(seed=3010, n=300)
    def sub(n):
        return (q["Tcw0"], q["points"][:n], q["obs"][:n], q["inv_sigma2"][:n])
    res = corb.Optimizer.PoseOptimizationBatch([sub(2), sub(8), sub(300), sub(0)], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    for T, out, ninl in (res[0], res[3]):
        assert np.array_equal(T, q["Tcw0"].reshape(4, 4)) and ninl == 0 and not out.any()
    def oracle(n, stages):
        e = np.zeros(n, pyorc.EDGE_DTYPE)
        e["pose"] = 0; e["point"] = np.arange(n); e["u"] = q["obs"][:n, 0]; e["v"] = q["obs"][:n, 1]; e["ur"] = q["obs"][:n, 2]; e["inv_sigma2"] = q["inv_sigma2"][:n]
        return pyorc.ba_solve_staged(q["Tcw0"].reshape(1, 16), np.zeros(1, np.uint8), q["points"][:n], np.ones(n, np.uint8), e, q["fx"], q["fy"], q["cx"], q["cy"], q["bf"], stages)
    r8 = oracle(8, pyorc.POSE_OPT_STAGES[:1])
    T, out, ninl = res[1]
    assert np.array_equal(out, r8["outlier"].astype(bool)) and ninl == 8 - int(r8["outlier"].sum()) and np.abs(T - r8["poses"][0]).max() < 1e-4
    r300 = oracle(300, pyorc.POSE_OPT_STAGES)
    T, out, ninl = res[2]
    assert np.array_equal(out, r300["outlier"].astype(bool)) and np.abs(T - r300["poses"][0]).max() < 1e-4
