"""Development aid: host / device phase times of one LocalBundleAdjustment call of the size the configs[2] replay produces (CORB_BA_TIMING=1)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
corb.warmup(0)
p = synth.local_ba_problem(seed=2100, n_local=5, n_fixed=4, pts_per_kf=550, outlier_frac=0.03)
a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
print("poses", len(p["poses"]), "points", len(p["points"]), "edges", len(p["edges"]))
for i in range(3):
    t0 = time.perf_counter(); g = corb.Optimizer.LocalBundleAdjustment(*a); dt = time.perf_counter() - t0
    print("call %d: %.2f ms wall, device %.2f ms, iters %d trials %d" % (i, dt * 1e3, g["ms_total"], g["iters_done"], g["trials"]), flush=True)
