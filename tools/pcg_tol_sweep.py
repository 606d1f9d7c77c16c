"""Development aid: effect of the PCG tolerance of the reduced camera system on the LM result and on the solve time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
p = synth.ba_problem(n_clients=8, kf_per_client=int(sys.argv[1]) if len(sys.argv) > 1 else 150, pts_per_kf=40, seed=1000, max_obs=8, window=6)
a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
ref = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, solver=2, pcg_tol=1e-13, pcg_max_iter=20000)
print("poses", len(p["poses"]), "edges", len(p["edges"]), "ref chi2", ref["chi2"][-1], "cg its", ref["pcg_iterations"], "solve ms", ref["ms"]["solve"])
for tol in (1e-10, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4):
    r = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, solver=2, pcg_tol=tol)
    dp = np.abs(r["poses"] - ref["poses"]).max() / max(1.0, np.abs(ref["poses"]).max())
    dx = np.abs(r["points"] - ref["points"]).max() / max(1.0, np.abs(ref["points"]).max())
    dc = max(abs(x - y) / y for x, y in zip(r["chi2"], ref["chi2"])) if len(r["chi2"]) == len(ref["chi2"]) else float("nan")
    print("tol %.0e  cg its %6d  solve ms %8.2f  total ms %8.2f  iters %d trials %d  rel dpose %.2e dpoint %.2e dchi2 %.2e" % (tol, r["pcg_iterations"], r["ms"]["solve"], r["ms"]["total"], r["iters_done"], r["trials"], dp, dx, dc))
