"""Development aid: where the wall time of the host-array global BA call goes at configs[4] size (CORB_BA_TIMING=1 prints the host phases), host flattening vs device flattening."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
kf = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
p = synth.ba_problem_fast(n_clients=8, kf_per_client=kf, pts_per_kf=100, seed=1000, obs_range=(3, 8), window=6)
a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
for dev in (False, True, False, True):
    t0 = time.time(); h = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=False, intr=p["intr"], devflat=dev); dt = time.time() - t0
    print("devflat %s: wall %.3f s device %.1f ms chi2 %.6e cg %d" % (dev, dt, h["ms"]["total"], h["chi2"][-1], h["pcg_iterations"]), file=sys.stderr, flush=True)
