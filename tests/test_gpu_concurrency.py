"""Thread safety of the host-pointer entry points: the reference calls its matchers and optimisers from several threads
(Tracking, LocalMapping, LoopClosing).  A long global BA (workspace lane 1) and per-frame matcher / pose calls (lane 0) run
concurrently from two threads and must return exactly what they return serially."""
import threading
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ba_and_tracking_calls_from_two_threads(corb, synth):
    prob = synth.ba_problem(n_clients=8, kf_per_client=60, pts_per_kf=40, seed=1007)
    a = (prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], prob["edges"], prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
    sc = synth.tracking_scene(4000)
    q = synth.pose_opt_problem(seed=3000, n=300)
    mt = corb.ORBmatcher(0.6, True)
    def track():
        m = mt.SearchByProjection_Frame(sc["cur"], sc["Tcw"], sc["Tlw"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["bf"], sc["mb"], sc["last"], sc["last_desc"], 7.0, False)
        p = corb.Optimizer.PoseOptimization(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
        return m, p
    ref_ba = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=5, solver=1)          # dense solver: run-to-run reproducible
    ref_m, ref_p = track()
    out = {}
    def ba_thread():
        out["ba"] = [corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=5, solver=1) for _ in range(3)]
    def track_thread():
        out["tr"] = [track() for _ in range(60)]
    t1 = threading.Thread(target=ba_thread); t2 = threading.Thread(target=track_thread)
    t1.start(); t2.start(); t1.join(); t2.join()
    for r in out["ba"]:
        assert np.allclose(r["chi2"], ref_ba["chi2"], rtol=1e-9) and np.abs(r["poses"] - ref_ba["poses"]).max() < 1e-6
    for m, p in out["tr"]:
        assert np.array_equal(m[0], ref_m[0]) and m[1] == ref_m[1]
        assert np.array_equal(p[1], ref_p[1]) and p[2] == ref_p[2] and np.abs(p[0] - ref_p[0]).max() < 1e-6
