"""CPU tests: staged optimisation of the oracle (LocalBundleAdjustment / PoseOptimization semantics)."""
import numpy as np


def _pose_opt(pyorc, q):
    n = len(q["points"])
    edges = np.zeros(n, pyorc.EDGE_DTYPE)
    edges["pose"] = 0; edges["point"] = np.arange(n); edges["u"] = q["obs"][:, 0]; edges["v"] = q["obs"][:, 1]; edges["ur"] = q["obs"][:, 2]
    edges["inv_sigma2"] = q["inv_sigma2"]
    return pyorc.ba_solve_staged(q["Tcw0"].reshape(1, 16), np.zeros(1, np.uint8), q["points"], np.ones(n, np.uint8), edges,
                                 q["fx"], q["fy"], q["cx"], q["cy"], q["bf"], pyorc.POSE_OPT_STAGES)


def test_pose_optimization_semantics(pyorc, synth):
    q = synth.pose_opt_problem()
    r = _pose_opt(pyorc, q)
    T = r["poses"][0]
    assert np.abs(T[:3, 3] - q["Tcw_true"][:3, 3]).max() < 0.02            # started 5-6 cm away
    assert np.abs(T[:3, :3] - q["Tcw_true"][:3, :3]).max() < 2e-3
    out = r["outlier"].astype(bool)
    # gross outliers are flagged, nearly all inliers kept (chi2 test at 95 %)
    assert out[q["outlier_truth"]].mean() > 0.95 and out[~q["outlier_truth"]].mean() < 0.12
    assert np.array_equal(r["points"], q["points"])                        # map points are fixed


def test_local_ba_semantics(pyorc, synth):
    p = synth.local_ba_problem()
    r = pyorc.ba_solve_staged(p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"],
                              p["fx"], p["fy"], p["cx"], p["cy"], p["bf"], pyorc.LOCAL_BA_STAGES)
    out = r["outlier"].astype(bool)
    assert out[p["outlier_truth"]].mean() > 0.9 and out[~p["outlier_truth"]].mean() < 0.1
    fixed = p["pose_fixed"].astype(bool)
    assert np.array_equal(r["poses"][fixed], p["poses"][fixed])
    # with (almost) noise-free inlier observations the free keyframes move towards the truth despite the outliers
    q = synth.local_ba_problem(seed=2005, pix_noise=0.05)
    rq = pyorc.ba_solve_staged(q["poses"], q["pose_fixed"], q["points"], q["point_fixed"], q["edges"],
                               q["fx"], q["fy"], q["cx"], q["cy"], q["bf"], pyorc.LOCAL_BA_STAGES)
    fq = q["pose_fixed"].astype(bool)
    err0 = np.abs(q["poses"][~fq][:, :3, 3] - q["poses_true"][~fq][:, :3, 3]).max()
    err1 = np.abs(rq["poses"][~fq][:, :3, 3] - q["poses_true"][~fq][:, :3, 3]).max()
    assert err1 < 0.5 * err0
    # a single robust stage without classification == plain robust BA with Huber sqrt(5.991)
    one = pyorc.ba_solve_staged(p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"],
                                p["fx"], p["fy"], p["cx"], p["cy"], p["bf"], [(5, 1, 1e30, 1e30, 0, 0, 0, 0, 0, pyorc._HM, pyorc._HS)])
    assert one["outlier"].sum() == 0 and one["iters_done"] <= 5


def test_local_ba_stop_flag_semantics(pyorc, synth):
    """pbStopFlag in Optimizer::LocalBundleAdjustment (Optimizer.cc:706-800): raised before the first optimize() -> plain return, nothing changes;
    raised during the first round -> the second round is skipped (bDoMore = false) but the final "Check inlier observations" pass (every edge,
    chi2 of its last computeError, fresh depth) and the write-back still happen -- the interrupted call equals ONE stage with the final test."""
    p = synth.local_ba_problem(seed=2011)
    a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    r0 = pyorc.ba_solve_staged(*a, pyorc.LOCAL_BA_STAGES, stop="before")
    assert np.array_equal(r0["poses"].reshape(-1, 16), p["poses"].reshape(-1, 16)) and np.array_equal(r0["points"], p["points"]) and r0["outlier"].sum() == 0 and r0["iters_done"] == 0
    r1 = pyorc.ba_solve_staged(*a, pyorc.LOCAL_BA_STAGES, stop="after_first_stage")
    one = pyorc.ba_solve_staged(*a, [pyorc.LOCAL_BA_STAGES[0][:6] + (1,) + pyorc.LOCAL_BA_STAGES[0][7:]])      # first round + the final (every-edge) test
    assert r1["iters_done"] == one["iters_done"] <= 5
    assert np.array_equal(r1["outlier"], one["outlier"]) and r1["outlier"].sum() > 0
    assert np.array_equal(r1["poses"], one["poses"]) and np.array_equal(r1["points"], one["points"])
    full = pyorc.ba_solve_staged(*a, pyorc.LOCAL_BA_STAGES)
    assert full["iters_done"] > r1["iters_done"]
