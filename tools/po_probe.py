"""Development aid: wall time of one PoseOptimization call of the size the configs[2] replay produces (about 1 750 observations)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
corb.warmup(0)
for n in (400, 1750):
    q = synth.pose_opt_problem(seed=3000, n=n)
    a = (q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    corb.Optimizer.PoseOptimization(*a)
    t0 = time.perf_counter()
    for _ in range(20): corb.Optimizer.PoseOptimization(*a)
    print("n %d: %.3f ms per call" % (n, (time.perf_counter() - t0) * 50))
