"""GPU parity: LocalBundleAdjustment / PoseOptimization through corb_ba_solve_staged vs the oracle (1e-4 relative,
identical outlier classification)."""
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
