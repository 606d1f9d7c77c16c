import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
prob = synth.ba_problem(n_clients=4, kf_per_client=25, pts_per_kf=30, seed=1004)
a = (prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], prob["edges"], prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
for solver in (1, 2):
    for it in (0, 1, 10):
        r = [corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=it, bRobust=False, solver=solver, devflat=df) for df in (False, False, True, True)]
        print("solver", solver, "iters", it, "host==host", np.array_equal(r[0]["chi2"], r[1]["chi2"]), "dev==dev", np.array_equal(r[2]["chi2"], r[3]["chi2"]),
              "host==dev", np.array_equal(r[0]["chi2"], r[2]["chi2"]), "poses", np.array_equal(r[0]["poses"], r[2]["poses"]), np.abs(r[0]["chi2"] - r[2]["chi2"]).max() / r[0]["chi2"][0],
              r[0]["structure"] == r[2]["structure"])
