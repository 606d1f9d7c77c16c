"""Development aid: how far the default PCG policy's result is from the oracle goldens (exact sparse LDL^T; 4 800 and 12 000 keyframes) -- run with CORB_BA_PCG_LOOSE=cap"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for name in ("ba_config3", "ba_12k"):
    gold = json.load(open(os.path.join(root, "tests", "golden", name + ".json")))
    kw = dict(gold["problem"]); kw["obs_range"] = tuple(kw["obs_range"]); kw["cams"] = [synth.KITTI_CAMS[c] for c in gold["cams"]]
    prob = synth.ba_problem_fast(**kw)
    a = (prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], prob["edges"], prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
    for tag, run in gold["runs"].items():
        g = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=10, bRobust=run["robust"], intr=prob["intr"])
        pi = np.asarray(gold["pose_sample"]); xi = np.asarray(gold["point_sample"])
        rp = np.asarray(run["poses"]).reshape(-1, 4, 4); rx = np.asarray(run["points"])
        same = g["iters_done"] == run["iters_done"] and g["trials"] == run["trials"]
        dchi = float(np.max(np.abs(np.asarray(g["chi2"]) / np.asarray(run["chi2"]) - 1))) if same else float("nan")
        dlam = float(np.max(np.abs(np.asarray(g["lam"]) / np.asarray(run["lam"]) - 1))) if same else float("nan")
        dt = float(np.abs(g["poses"][pi][:, :3, 3] - rp[:, :3, 3]).max() / max(1.0, np.abs(rp[:, :3, 3]).max()))
        dR = float(np.abs(g["poses"][pi][:, :3, :3] - rp[:, :3, :3]).max())
        dx = float(np.abs(g["points"][xi] - rx).max() / max(1.0, np.abs(rx).max()))
        print("%s %s: counts equal %s  cg %d  refined %s  residual_max %.2e | rel dchi2 %.2e  dlambda %.2e  dt %.2e  dR %.2e  dpoint %.2e  (bars 1e-4, 1e-3, 1e-4, 1e-4, 1e-4)" % (
            name, tag, same, g["pcg_iterations"], g["certificate"]["pcg_refined_trials"], g["certificate"]["pcg_residual_max"], dchi, dlam, dt, dR, dx))
