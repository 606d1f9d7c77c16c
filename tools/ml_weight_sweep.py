"""Development aid: CG iterations / solve time of the global BA against the weight of the multilevel preconditioner's coarse terms (CORB_BA_ML_W, corb_ba.cpp
ml_level_weight) at several map sizes.  usage: ml_weight_sweep.py [kf_per_client ...]   (8 clients, 100 points per keyframe, 3..8 observations)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
sizes = [int(a) for a in sys.argv[1:]] or [150, 600, 2500, 6250]
weights = os.environ.get("ML_WEIGHTS", "1;0.5;0.35;0.25;0.15;0.35,0.2,0;0.35,0.2,0.1;0.5,0.25,0.12,0.06").split(";")
for kf in sizes:
    p = synth.ba_problem_fast(n_clients=8, kf_per_client=kf, pts_per_kf=100, seed=1000, obs_range=(3, 8), window=6)
    a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    for rob in (False, True):
        ref = None
        for w in weights:
            os.environ["CORB_BA_ML_W"] = w
            g = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=rob, intr=p["intr"])
            g = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=rob, intr=p["intr"])
            if ref is None: ref = g
            print("kf %6d robust %d w %-22s cg %5d  solve %7.2f ms  total %7.2f ms  trials %2d  chi2 rel diff %.1e  residual max %.1e  levels %d" % (
                8 * kf, rob, w, g["pcg_iterations"], g["ms"]["solve"], g["ms"]["total"], g["trials"], abs(g["chi2"][-1] - ref["chi2"][-1]) / ref["chi2"][-1],
                g["certificate"]["pcg_residual_max"], g["structure"].get("pc_levels", 0)), flush=True)
