"""Development aid: global BA from store records vs from host arrays at one size (8 clients x KF keyframes, 100 points per keyframe, 3..8 observations)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
kf = int(sys.argv[1]) if len(sys.argv) > 1 else 150
t0 = time.time(); p = synth.ba_problem_fast(n_clients=8, kf_per_client=kf, pts_per_kf=100, seed=1000, obs_range=(3, 8), window=6); tg = time.time() - t0
a = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
for _ in range(2):
    t0 = time.time(); h = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=False, intr=p["intr"]); dth = time.time() - t0
t0 = time.time(); ma = synth.map_arrays(p, kf, 100)
KF = corb.KeyFrameStore(len(p["poses"]), ma["max_features"]); MP = corb.MapPointStore(len(p["points"]), ma["max_obs"])
KF.put_batch(0, ma["meta"], ma["feat_off"], ma["kp"], None, ma["ur"], None, ma["mp_id"]); MP.put(0, ma["mp_records"], ma["obs_off"], ma["obs_kf"], ma["obs_idx"]); ts = time.time() - t0
ks = np.arange(len(p["poses"]), dtype=np.int32); ms = np.arange(len(p["points"]), dtype=np.int32)
for _ in range(2):
    t0 = time.time(); g = corb.GlobalBundleAdjustemntStore(KF, ks, MP, ms, nIterations=10, bRobust=False, nLoopKF=7, fetch=False); dtg = time.time() - t0
if os.environ.get("BA_REPEATS"):
    tot, sol = [], []
    for _ in range(int(os.environ["BA_REPEATS"])):
        g = corb.GlobalBundleAdjustemntStore(KF, ks, MP, ms, nIterations=10, bRobust=False, nLoopKF=7, fetch=False); tot.append(g["ms"]["total"]); sol.append(g["ms"]["solve"])
    print("repeats: device total min %.2f median %.2f | solve min %.2f median %.2f ms" % (min(tot), sorted(tot)[len(tot) // 2], min(sol), sorted(sol)[len(sol) // 2]))
print("poses %d points %d edges %d | gen %.1fs staging %.1fs | host arrays: wall %.3fs device %.1f ms | store: wall %.3fs device %.1f ms | chi2 %.6e vs %.6e equal %s | %s" % (
    len(p["poses"]), len(p["points"]), len(p["edges"]), tg, ts, dth, h["ms"]["total"], dtg, g["ms"]["total"], h["chi2"][-1], g["chi2"][-1], np.array_equal(h["chi2"], g["chi2"]), g["ms"]))
print("host arrays: cg %d trials %d %s lam %s | store: cg %d trials %d %s" % (h["pcg_iterations"], h["trials"], h["certificate"], np.array2string(h["lam"], precision=4), g["pcg_iterations"], g["trials"], g["certificate"]))
