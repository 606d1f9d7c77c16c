import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
for kw in (dict(), dict(n_local=16, n_fixed=24, pts_per_kf=100)):
    p = synth.local_ba_problem(seed=2000, **kw)
    la = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    corb.Optimizer.LocalBundleAdjustment(*la)
    t0 = time.perf_counter(); r = corb.Optimizer.LocalBundleAdjustment(*la); dt = time.perf_counter() - t0
    print("wall ms", dt * 1e3, "iters", r["iters_done"], "trials", r["trials"], "device ms", r.get("ms"))
