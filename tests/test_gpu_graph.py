"""GPU parity of corb_optimize_essential_graph vs the oracle: 1e-4 relative on the optimised similarities, poses and points.
chi2 per iteration is compared at 1e-3 of the initial chi2: g2o differentiates EdgeSim3 numerically (delta 1e-9) and starts
Levenberg at lambda = 1e-16, so the Gauss-Newton steps amplify the rounding noise of the Jacobians (different libm on CPU and
GPU) -- late iterations agree to ~1e-4 of the initial chi2, the estimates to ~1e-5 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.mark.parametrize("seed,K,fix_scale", [(7100, 60, False), (7101, 100, False), (7102, 80, True), (7103, 150, False)])
def test_matches_oracle(corb, pyorc, synth, seed, K, fix_scale):
    g = synth.essential_graph(seed, K=K)
    G = corb.Optimizer.OptimizeEssentialGraph(g, 20, fix_scale)
    R = pyorc.optimize_essential_graph(g, 20, fix_scale)
    n = min(len(G["chi2"]), len(R["chi2"]))
    assert np.allclose(G["chi2"][:n], R["chi2"][:n], rtol=0, atol=1e-3 * R["chi2"][0])
    assert np.allclose(G["chi2"][:2], R["chi2"][:2], rtol=1e-4)                 # the first step is far above the noise
    assert abs(G["chi2"][-1] - R["chi2"][-1]) <= 1e-3 * R["chi2"][0]
    assert np.abs(G["S"] - R["S"]).max() <= RTOL * max(1.0, np.abs(R["S"]).max())
    assert np.abs(G["Tiw"] - R["Tiw"]).max() <= RTOL * max(1.0, np.abs(R["Tiw"]).max())
    assert np.abs(G["points"] - R["points"]).max() <= RTOL * max(1.0, np.abs(R["points"]).max())
    assert np.array_equal(G["S"][0], g["S"][0])
    if fix_scale:
        assert np.array_equal(G["S"][:, 7], g["S"][:, 7])
    # Why chi2 is held to 1e-3 chi2_0 and not to 1e-4 relative: the ORACLE ITSELF moves by that much when one measurement is perturbed by one unit in the last
    # place (lambda = 1e-16 makes every step a Gauss-Newton step on numerically differentiated Jacobians: rounding noise of 1e-16 / delta 1e-9 = 1e-7 per Jacobian
    # entry is amplified by the conditioning of the 7 K-dimensional normal equations).  The GPU result has to sit inside a few multiples of that sensitivity.
    if K <= 100:
        g2 = dict(g); m2 = np.array(g["meas"], np.float64).copy(); m2[0, 4] = np.nextafter(m2[0, 4], np.inf); g2["meas"] = m2
        R2 = pyorc.optimize_essential_graph(g2, 20, fix_scale)
        n2 = min(n, len(R2["chi2"]))
        own = np.abs(R2["chi2"][:n2] - R["chi2"][:n2]).max()                      # the oracle's own sensitivity to a 1-ulp input change
        assert np.abs(G["chi2"][:n2] - R["chi2"][:n2]).max() <= max(50.0 * own, 1e-6 * R["chi2"][0]), (own, np.abs(G["chi2"][:n2] - R["chi2"][:n2]).max())


def test_repeated_runs_are_bit_identical(corb, synth):
    """the normal equations are accumulated without atomics (per-vertex / per-pair gathers in edge order): with lambda = 1e-16 and numeric Jacobians any
    run-to-run rounding difference would be amplified into visibly different late iterations"""
    g = synth.essential_graph(7101, K=100)
    a = corb.Optimizer.OptimizeEssentialGraph(g, 20, False)
    for _ in range(4):
        b = corb.Optimizer.OptimizeEssentialGraph(g, 20, False)
        assert np.array_equal(a["chi2"], b["chi2"]) and np.array_equal(a["S"], b["S"]) and np.array_equal(a["points"], b["points"])


def test_all_fixed_and_no_edges(corb, synth):
    g = synth.essential_graph(7110, K=20)
    g2 = dict(g); g2["fixed"] = np.ones(20, np.uint8)
    G = corb.Optimizer.OptimizeEssentialGraph(g2, 20, False)
    assert G["iters_done"] == 0 and np.array_equal(G["S"], g["S"])
