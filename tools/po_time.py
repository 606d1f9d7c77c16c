import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import time, numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
q = synth.pose_opt_problem(seed=3000, n=400)
a = (q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
for solver in (3, 1):
    corb.Optimizer.PoseOptimization(*a, solver=solver)
    t0 = time.perf_counter()
    for _ in range(10): corb.Optimizer.PoseOptimization(*a, solver=solver)
    print("solver", solver, "ms/call", (time.perf_counter() - t0) * 100)
qs = [synth.pose_opt_problem(seed=3100 + i, n=400) for i in range(64)]
frames = [(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"]) for q in qs]
for nb in (1, 8, 64, 512):
    fr = (frames * ((nb + 63) // 64))[:nb]
    corb.Optimizer.PoseOptimizationBatch(fr, q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    t0 = time.perf_counter()
    for _ in range(3): corb.Optimizer.PoseOptimizationBatch(fr, q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    dt = (time.perf_counter() - t0) / 3
    print("batch", nb, "ms", dt * 1e3, "frames/s", nb / dt)
