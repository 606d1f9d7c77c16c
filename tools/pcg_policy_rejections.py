"""Development aid: the default PCG policy (cap BA_PCG_TOL_LOOSE, or CORB_BA_PCG_LOOSE) against the dense solver on noisy maps above 256 keyframes whose LM runs reject trials:
the accept / reject history, chi2 per iteration, lambda"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
worst = dict(chi=0.0, lam=0.0, pose=0.0); bad = 0; rej = 0
for seed in range(4100, 4100 + (int(sys.argv[1]) if len(sys.argv) > 1 else 12)):
    for robust in (False, True):
        p = synth.ba_problem(n_clients=8, kf_per_client=36 + seed % 5, pts_per_kf=24, seed=seed, pose_noise=(0.5, 0.08), point_noise=0.6 if seed & 1 else 0.25)
        a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
        d = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=robust, solver=1)
        g = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=robust, solver=2)
        same = d["iters_done"] == g["iters_done"] and d["trials"] == g["trials"]
        rej += d["trials"] - d["iters_done"]
        if not same:
            bad += 1; print("seed", seed, robust, "history differs: dense", d["iters_done"], d["trials"], "pcg", g["iters_done"], g["trials"], "refined", g["certificate"]["pcg_refined_trials"]); continue
        dc = float(np.max(np.abs(np.asarray(g["chi2"]) / np.asarray(d["chi2"]) - 1))); dl = float(np.max(np.abs(np.asarray(g["lam"]) / np.asarray(d["lam"]) - 1)))
        dp = float(np.abs(g["poses"] - d["poses"]).max())
        worst["chi"] = max(worst["chi"], dc); worst["lam"] = max(worst["lam"], dl); worst["pose"] = max(worst["pose"], dp)
        print("seed", seed, "robust", int(robust), "poses", len(p["poses"]), "trials", d["trials"], "iters", d["iters_done"], "refined", int(g["certificate"]["pcg_refined_trials"]), "dchi2 %.1e dlam %.1e dpose %.1e" % (dc, dl, dp))
print("histories that differ", bad, "| rejected trials seen", rej, "| worst", worst)
