"""Development aid: dense Cholesky (solver 1) against PCG with 16-keyframe blocks + coarse levels (solver 2) on small maps, device ms per 10 LM iterations."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import corbload; corb = corbload.load_pkg()
from corb_slam_amd import synth
for kf in (64, 100, 128, 160, 200, 256):
    p = synth.ba_problem_fast(n_clients=1, kf_per_client=kf, pts_per_kf=100, seed=1000, obs_range=(3, 8), window=6)
    a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    out = []
    for sv in (1, 2):
        for rep in range(2):
            r = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=False, solver=sv, intr=p["intr"])
        out.append("solver %d: %.1f ms (solve %.1f, cg %d, trials %d, chi2 %.10e)" % (sv, r["ms"]["total"], r["ms"]["solve"], r["pcg_iterations"], r["trials"], r["chi2"][-1]))
    print(kf, " | ".join(out), flush=True)
