"""Development aid: the other optimisers -- LocalBundleAdjustment (small and 24-keyframe windows: one-workgroup, chained and dense-solver routes), OptimizeEssentialGraph
(dense Cholesky), a mid-size global BA on the fused path -- on one thread beside tracking calls and stereo front-end runs on another: every call must return its serial bits"""
import sys, os, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
def lba_args(p): return (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
jobs = {}
for name, kw in (("lba 6+4", dict(seed=2001)), ("lba 24+8", dict(seed=2002, n_local=24, n_fixed=8, pts_per_kf=40)), ("lba 40+10", dict(seed=2003, n_local=40, n_fixed=10, pts_per_kf=30))):
    p = synth.local_ba_problem(**kw); jobs[name] = (lambda p=p: corb.Optimizer.LocalBundleAdjustment(*lba_args(p)), ("poses", "points", "outlier"))
g = synth.essential_graph(seed=7001, K=80)
jobs["essential graph 80"] = (lambda: corb.Optimizer.OptimizeEssentialGraph(g, iterations=12), ("S", "chi2", "points"))
pm = synth.ba_problem(n_clients=3, kf_per_client=30, pts_per_kf=30, seed=1013)
jobs["global BA 90 KF (dense)"] = (lambda: corb.Optimizer.GlobalBundleAdjustemnt(*lba_args(pm), nIterations=6, bRobust=True, solver=1), ("chi2", "poses", "points"))
sc = synth.tracking_scene(4000); q = synth.pose_opt_problem(seed=3000, n=900); mt = corb.ORBmatcher(0.6, True)
W, H, NB = 1241, 376, 4
fr = [synth.stereo_pair(i, w=W, h=H) for i in range(70, 70 + NB)]; P = np.ascontiguousarray(np.stack([np.stack([l, r]) for l, r in fr]))
sf = corb.StereoFrontend(nfeatures=2000, width=W, height=H, max_frames=NB)
def other():
    sf.upload_batch(0, P); sf.run(NB); sf.sync()
    mt.SearchByProjection_Frame(sc["cur"], sc["Tcw"], sc["Tlw"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["bf"], sc["mb"], sc["last"], sc["last_desc"], 7.0, False)
    corb.Optimizer.PoseOptimization(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
def same(a, b, keys): return all(np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes() for k in keys)
refs = dict((n, f()) for n, (f, _) in jobs.items())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
stop = [False]
def bg():
    while not stop[0]: other()
t = threading.Thread(target=bg); t.start()
bad = dict((n, 0) for n in jobs)
for rep in range(N):
    for n, (f, keys) in jobs.items():
        if not same(f(), refs[n], keys): bad[n] += 1
stop[0] = True; t.join()
for n in jobs: print("%-28s deviating %d of %d" % (n, bad[n], N))
