import sys, os, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/corbload.py') else '.')
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
from oracle import pyorc
def A(p): return (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
p = synth.ba_problem_fast(n_clients=4, kf_per_client=40, pts_per_kf=30, seed=1020)
r = pyorc.ba_solve(*A(p), iters=10, robust=False)
for solver in (1, 2):
    g = corb.Optimizer.GlobalBundleAdjustemnt(*A(p), nIterations=10, bRobust=False, solver=solver)
    g2 = corb.Optimizer.GlobalBundleAdjustemnt(*A(p), nIterations=10, bRobust=False, solver=solver)
    print("solver", solver, "rel chi2 err", np.abs(g["chi2"] - r["chi2"]).max() / r["chi2"][0], np.allclose(g["chi2"], r["chi2"], rtol=1e-4), "identical runs:", np.array_equal(g["chi2"], g2["chi2"]) and np.array_equal(g["poses"], g2["poses"]) and np.array_equal(g["points"], g2["points"]), g["iters_done"], r["iters_done"], g["trials"], r["trials"])
for kf in (150, 1250):
    p = synth.ba_problem_fast(n_clients=8, kf_per_client=kf, pts_per_kf=40, seed=1000, obs_range=(8, 8))
    for env in ("", "1"):
        if env: os.environ["CORB_BA_ATOMIC_SCHUR"] = "1"
        else: os.environ.pop("CORB_BA_ATOMIC_SCHUR", None)
        corb.Optimizer.GlobalBundleAdjustemnt(*A(p), nIterations=2, bRobust=False, solver=2)
        t0 = time.time(); g = corb.Optimizer.GlobalBundleAdjustemnt(*A(p), nIterations=5, bRobust=False, solver=2); dt = time.time() - t0
        print("kf", kf, "atomic" if env else "mfma", "edges", len(p["edges"]), "wall %.3f" % dt, g["ms"], "cg", g["pcg_iterations"], g["chi2"][[0, -1]], g["trials"])
