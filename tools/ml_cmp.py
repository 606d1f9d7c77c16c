import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import corbload; corb=corbload.load_pkg()
from corb_slam_amd import synth
for nc, kf in ((1, 6250), (2, 3000), (8, 1200)):
    p = synth.ba_problem_fast(n_clients=nc, kf_per_client=kf, pts_per_kf=100, seed=1000, obs_range=(3, 8), window=6)
    a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    for ml in (1, 2):
        tot = []
        for it in (1, 4, 7, 10):
            r = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=it, bRobust=False, solver=2, intr=p["intr"], pc_multilevel=ml)
            tot.append(r["pcg_iterations"])
        print(nc, kf, "ml" if ml == 2 else "bj", "levels", r["structure"]["pc_levels"], "cumulative cg after 1/4/7/10 LM its", tot, "solve ms %.1f" % r["ms"]["solve"], "chi2 %.6e" % r["chi2"][-1], flush=True)
