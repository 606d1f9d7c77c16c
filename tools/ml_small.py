"""Development aid: the reduced solve of small and mid-size maps with and without the multilevel preconditioner (device ms per 10 LM iterations)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import corbload; corb = corbload.load_pkg()
from corb_slam_amd import synth
for nc, kf in ((1, 280), (1, 400), (1, 600), (1, 1200), (2, 800), (1, 2000)):
    p = synth.ba_problem_fast(n_clients=nc, kf_per_client=kf, pts_per_kf=100, seed=1000, obs_range=(3, 8), window=6)
    a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    for name, kw in (("auto", dict()), ("bj16", dict(pc_block=16, pc_multilevel=1)), ("ml", dict(pc_block=16, pc_multilevel=2))):
        for rep in range(2):
            r = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=False, solver=2, intr=p["intr"], **kw)
        print(nc, kf, name, "levels", r["structure"]["pc_levels"], "block", r["structure"]["pc_block"], "cg", r["pcg_iterations"], "ms total %.1f solve %.1f schur %.1f" % (r["ms"]["total"], r["ms"]["solve"], r["ms"]["schur"]), "chi2 %.9e" % r["chi2"][-1], flush=True)
