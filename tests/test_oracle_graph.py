"""Oracle sanity for Optimizer::OptimizeEssentialGraph: the Sim3 log/exp pair is consistent, the optimisation closes the loop
(chi2 drops by orders of magnitude, trajectory error shrinks), fixed vertices and bFixScale are honoured, the SE3 recovery and
the map point correction follow their definitions."""
import numpy as np
import pytest


def _centres(T):
    return np.array([-t[:3, :3].T @ t[:3, 3] for t in T])


@pytest.mark.parametrize("seed,K", [(7000, 60), (7001, 90)])
def test_closes_the_loop(pyorc, synth, seed, K):
    g = synth.essential_graph(seed, K=K)
    r = pyorc.optimize_essential_graph(g, 20, False)
    assert r["chi2"][-1] < 0.05 * r["chi2"][0] and r["iters_done"] >= 2
    assert np.all(np.diff(r["chi2"]) <= 1e-9)                                   # LM never accepts an increase
    ct = _centres(g["Ttrue"]); before = np.linalg.norm(_centres(g["Test"]) - ct, axis=1).mean()
    after = np.linalg.norm(_centres(r["Tiw"].astype(np.float64)) - ct, axis=1).mean()
    assert after < 0.35 * before
    assert np.array_equal(r["S"][0], g["S"][0])                                 # the loop keyframe is fixed


def test_fix_scale_and_apply(pyorc, synth):
    g = synth.essential_graph(7002, K=50)
    r = pyorc.optimize_essential_graph(g, 20, True)
    assert np.array_equal(r["S"][:, 7], g["S"][:, 7])                           # scales untouched
    # SE3 recovery: Tiw = [R | t / s]
    from scipy.spatial.transform import Rotation
    k = 17
    R = Rotation.from_quat(r["S"][k, :4]).as_matrix()
    assert np.allclose(r["Tiw"][k][:3, :3], R, atol=1e-6) and np.allclose(r["Tiw"][k][:3, 3], r["S"][k, 4:7] / r["S"][k, 7], atol=1e-5)
    # map point correction: Swr_new(Srw_old(p)); untouched without a reference keyframe
    m = int(np.nonzero(g["ref"] >= 0)[0][0]); ref = g["ref"][m]
    a = g["S"][ref, 7] * Rotation.from_quat(g["S"][ref, :4]).apply(g["points"][m].astype(np.float64)) + g["S"][ref, 4:7]
    Sn = synth.sim3_inv(r["S"][ref]); c = Sn[7] * Rotation.from_quat(Sn[:4]).apply(a) + Sn[4:7]
    assert np.allclose(r["points"][m], c, atol=1e-4)
    assert np.array_equal(r["points"][g["ref"] < 0], g["points"][g["ref"] < 0])
