"""Development aid: random small / mid-size bundle adjustments (all three solver routes, robust or not, random fixed vertices, odd sizes) against the CPU oracle at the parity bar"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
from oracle import pyorc
pyorc.build()
import busy; busy.start(corb, synth)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
bad = 0
for it in range(N):
    nc = int(rng.integers(1, 6)); kf = int(rng.integers(3, 40)); pts = int(rng.integers(5, 45)); robust = bool(it & 1); solver = int(rng.choice([0, 1, 2]))
    p = synth.ba_problem(n_clients=nc, kf_per_client=kf, pts_per_kf=pts, seed=7000 + it, window=int(rng.integers(2, 8)))
    nfix = int(rng.integers(0, 4))
    if nfix: p["point_fixed"][rng.choice(len(p["points"]), size=min(nfix, len(p["points"])), replace=False)] = 1
    if it % 5 == 0 and len(p["poses"]) > 3: p["pose_fixed"][int(rng.integers(1, len(p["poses"])))] = 1
    a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    g = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=robust, solver=solver)
    r = pyorc.ba_solve(*a, iters=10, robust=robust)
    ok = g["iters_done"] == r["iters_done"] and g["trials"] == r["trials"] and np.allclose(g["chi2"], r["chi2"], rtol=1e-4) and \
        np.abs(g["poses"] - r["poses"]).max() <= 1e-4 * max(1.0, np.abs(r["poses"]).max()) and np.abs(g["points"] - r["points"]).max() <= 1e-4 * max(1.0, np.abs(r["points"]).max())
    if not ok:
        bad += 1
        print("case", it, "clients", nc, "kf", kf, "pts", pts, "robust", robust, "solver", solver, "iters", g["iters_done"], r["iters_done"], "trials", g["trials"], r["trials"],
              "dchi2", float(np.max(np.abs(np.asarray(g["chi2"])[:min(len(g["chi2"]), len(r["chi2"]))] / np.asarray(r["chi2"])[:min(len(g["chi2"]), len(r["chi2"]))] - 1))))
print("cases", N, "outside the bar", bad)
