"""GPU box: block-Jacobi block size sweep of the PCG solver on the bench's 1 200-keyframe map (and a 10 000-keyframe map)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
for kf in (150, 1250):
    prob = synth.ba_problem(n_clients=8, kf_per_client=kf, pts_per_kf=40, seed=1000, max_obs=8, window=6)
    args = (prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], prob["edges"], prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
    corb.Optimizer.GlobalBundleAdjustemnt(*args, nIterations=2, bRobust=False, solver=2)
    for pc in (1, 8, 16, 32, 64):
        t0 = time.perf_counter()
        g = corb.Optimizer.GlobalBundleAdjustemnt(*args, nIterations=10 if kf == 150 else 5, bRobust=False, solver=2, pc_block=pc)
        dt = time.perf_counter() - t0
        print("poses %d pc_block %2d: wall %.1f ms, device %s, cg its %d, chi2 %.6e" % (len(prob["poses"]), pc, dt * 1e3,
              {k: round(v, 1) for k, v in g["ms"].items()}, g["pcg_iterations"], g["chi2"][-1]), flush=True)
