"""Development aid: global BA wall / device time by map size (8 clients, block-sparse PCG)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
PTS = 40
argv = sys.argv[1:]
if argv and argv[0] == "--pts":
    PTS = int(argv[1]); argv = argv[2:]          # 100 per keyframe = the 5 M points of BASELINE config 5 at 50 000 keyframes
OBS = (8, 8)                                      # 8 observations per point, like round 1's profiles
if argv and argv[0] == "--obs":
    OBS = (int(argv[1]), int(argv[2])); argv = argv[3:]      # 3 8 = SURVEY s8d(ii): the 3..8 nearest keyframes, mean 5.5
ITERS = 5
if argv and argv[0] == "--iters":
    ITERS = int(argv[1]); argv = argv[2:]
for kf in [int(a) for a in argv] or [150, 600]:
    t0 = time.time()
    p = synth.ba_problem_fast(n_clients=8, kf_per_client=kf, pts_per_kf=PTS, seed=1000, obs_range=OBS, window=6)
    tg = time.time() - t0
    a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    t0 = time.time()
    r = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=ITERS, bRobust=False, solver=2, intr=p["intr"])
    dt_cold = time.time() - t0                  # first call at this size: the host staging vectors grow
    t0 = time.time()
    r = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=ITERS, bRobust=False, solver=2, intr=p["intr"])
    dt = time.time() - t0
    print("poses %6d points %8d edges %9d | gen %.1fs | wall %.2fs (first call %.2fs) device %.1f ms (build %.1f schur %.1f solve %.1f) cg %d iters %d trials %d chi2 %.4e -> %.4e" % (
        len(p["poses"]), len(p["points"]), len(p["edges"]), tg, dt, dt_cold, r["ms"]["total"], r["ms"]["build"], r["ms"]["schur"], r["ms"]["solve"], r["pcg_iterations"], r["iters_done"], r["trials"], r["chi2"][0], r["chi2"][-1]), flush=True)
