"""Development aid: a dense-solver global BA on one thread beside tracking calls on another, many times: what deviates from the serial result, and by how much"""
import sys, os, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
prob = synth.ba_problem(n_clients=8, kf_per_client=60, pts_per_kf=40, seed=1007)
a = (prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], prob["edges"], prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
sc = synth.tracking_scene(4000)
q = synth.pose_opt_problem(seed=3000, n=300)
mt = corb.ORBmatcher(0.6, True)
def track():
    m = mt.SearchByProjection_Frame(sc["cur"], sc["Tcw"], sc["Tlw"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["bf"], sc["mb"], sc["last"], sc["last_desc"], 7.0, False)
    p = corb.Optimizer.PoseOptimization(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
    return m, p
ref = corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=5, solver=1)
ref_m, ref_p = track()
nbad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    out = {}
    def ba_thread(): out["ba"] = [corb.Optimizer.GlobalBundleAdjustemnt(*a, nIterations=5, solver=1) for _ in range(3)]
    def track_thread(): out["tr"] = [track() for _ in range(60)]
    t1 = threading.Thread(target=ba_thread); t2 = threading.Thread(target=track_thread)
    t1.start(); t2.start(); t1.join(); t2.join()
    for k, r in enumerate(out["ba"]):
        if not np.array_equal(r["chi2"], ref["chi2"]) or r["poses"].tobytes() != ref["poses"].tobytes():
            nbad += 1
            print("rep", rep, "call", k, "chi2", r["chi2"], "ref", ref["chi2"], "lam", r["lam"], ref["lam"], "trials", r["trials"], ref["trials"], "poses", np.abs(r["poses"] - ref["poses"]).max())
    tb = sum(1 for m, p in out["tr"] if not (np.array_equal(m[0], ref_m[0]) and np.array_equal(p[1], ref_p[1]) and p[0].tobytes() == ref_p[0].tobytes()))
    if tb: print("rep", rep, "tracking mismatches", tb)
print("deviating BA calls", nbad)
