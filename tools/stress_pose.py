"""Development aid: random PoseOptimization problems (sizes 60 .. 3 000 observations) against the oracle: outlier flags equal, pose at the parity bar"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
from oracle import pyorc
pyorc.build()
import busy; busy.start(corb, synth)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(515)
bad = flags = 0
for it in range(N):
    n = int(rng.choice([60, 130, 257, 400, 511, 640, 1000, 1750, 2049, 3000]))
    q = synth.pose_opt_problem(seed=9000 + it, n=n)
    T, outl, ninl = corb.Optimizer.PoseOptimization(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    edges = np.zeros(n, pyorc.EDGE_DTYPE)
    edges["pose"] = 0; edges["point"] = np.arange(n); edges["u"] = q["obs"][:, 0]; edges["v"] = q["obs"][:, 1]; edges["ur"] = q["obs"][:, 2]; edges["inv_sigma2"] = q["inv_sigma2"]
    r = pyorc.ba_solve_staged(q["Tcw0"].reshape(1, 16), np.zeros(1, np.uint8), q["points"], np.ones(n, np.uint8), edges, q["fx"], q["fy"], q["cx"], q["cy"], q["bf"], pyorc.POSE_OPT_STAGES)
    same_flags = np.array_equal(outl, r["outlier"].astype(bool))
    dev = float(np.abs(T - r["poses"][0]).max() / max(1.0, np.abs(r["poses"][0]).max()))
    if not same_flags: flags += 1
    if not same_flags or dev > 1e-4: bad += 1; print("case", it, "n", n, "flags equal", same_flags, "differing", int((outl != r["outlier"].astype(bool)).sum()), "pose dev %.2e" % dev)
print("cases", N, "outside the bar", bad, "| with differing flags", flags)
