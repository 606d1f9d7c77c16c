"""Development aid: dense Cholesky (rocSOLVER) vs block-sparse PCG for the reduced camera system by map size."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
for kf in (5, 10, 20, 40, 64, 100):
    p = synth.ba_problem(n_clients=8, kf_per_client=kf, pts_per_kf=40, seed=1000, max_obs=8, window=6)
    a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    out = []
    for solver in (1, 2):
        corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, solver=solver)
        r = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, solver=solver)
        out.append("solver %d: total %7.2f ms solve %7.2f ms (cg %5d) chi2 %.6e" % (solver, r["ms"]["total"], r["ms"]["solve"], r["pcg_iterations"], r["chi2"][-1]))
    print("poses %4d edges %6d | %s | %s" % (len(p["poses"]), len(p["edges"]), out[0], out[1]))
