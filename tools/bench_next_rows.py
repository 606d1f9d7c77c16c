"""Measurement of the SURVEY §8f ("next") rows on the GPU box: wall time per call through the C-ABI (host pointers in, results out,
i.e. INCLUDING uploads / downloads / allocation) beside the CPU oracle (1 thread) on the same synthetic inputs.
Usage: python tools/bench_next_rows.py > profiles/r01_next_rows.json"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
import pyorc


def timeit(f, reps=10):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return round(1e3 * float(np.median(ts)), 4)


rows = []
def row(name, ref, units, gpu, cpu, reps=10, cpu_reps=3):
    g = timeit(gpu, reps); c = timeit(cpu, cpu_reps)
    rows.append(dict(routine=name, reference=ref, workload=units, gpu_ms_per_call=g, cpu_oracle_ms_per_call=c, speedup=round(c / g, 2)))

mt = corb.ORBmatcher(0.6, True)
sc = synth.tracking_scene(4000)
row("SearchByProjection(Frame, MapPoints)", "ORBmatcher.cc:45-131", "2000 features x 2000 map points",
    lambda: mt.SearchByProjection(sc["cur"], sc["mps"], sc["last_desc"], 3.0), lambda: pyorc.search_by_projection_map(sc["cur"], sc["mps"], sc["last_desc"], 3.0, 0.6))
row("SearchByProjection(Frame, LastFrame)", "ORBmatcher.cc:1470-1614", "2000 x 2000",
    lambda: mt.SearchByProjection_Frame(sc["cur"], sc["Tcw"], sc["Tlw"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["bf"], sc["mb"], sc["last"], sc["last_desc"], 7.0, False),
    lambda: pyorc.search_by_projection_frame(sc["cur"], sc["Tcw"], sc["Tlw"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["bf"], sc["mb"], sc["last"], sc["last_desc"], 7.0, 0, 1))
kf = synth.keyframe_scene(5000)
row("SearchByProjection(Frame, KeyFrame) relocalisation", "ORBmatcher.cc:1616-1744", "2000 x 2000",
    lambda: mt.SearchByProjection_Reloc(kf["kf2"], kf["claimed2"], kf["T2w"], kf["pts1"], kf["desc1"], 10.0, 100),
    lambda: pyorc.search_by_projection_reloc(kf["kf2"], kf["claimed2"], kf["T2w"], kf["pts1"], kf["desc1"], 10.0, 100, 1))
row("Fuse(KeyFrame, MapPoints)", "ORBmatcher.cc:960-1116", "2000 x 2000",
    lambda: mt.Fuse(kf["kf2"], kf["T2w"], kf["Ow2"], kf["pts1"], kf["desc1"], 3.0), lambda: pyorc.fuse(kf["kf2"], kf["T2w"], kf["Ow2"], 0, kf["pts1"], kf["desc1"], 3.0))
a3 = (kf["kf1"], kf["kf2"], kf["T1w"], kf["T2w"], kf["pts1"], kf["desc1"], kf["pts2"], kf["desc2"], kf["s12"], kf["R12"], kf["t12"], 7.5)
row("SearchBySim3", "ORBmatcher.cc:1244-1468", "2000 x 2000, both directions", lambda: mt.SearchBySim3(*a3), lambda: pyorc.search_by_sim3(*a3))

for kw in (dict(), dict(n_local=16, n_fixed=24, pts_per_kf=100)):
    p = synth.local_ba_problem(seed=2000, **kw)
    la = (p["poses"], p["pose_fixed"], p["points"], p["point_fixed"], p["edges"], p["fx"], p["fy"], p["cx"], p["cy"], p["bf"])
    row("LocalBundleAdjustment", "Optimizer.cc:487-838", "%d keyframes (%d free), %d points, %d observations" % (len(p["poses"]), int((p["pose_fixed"] == 0).sum()), len(p["points"]), len(p["edges"])),
        lambda: corb.Optimizer.LocalBundleAdjustment(*la), lambda: pyorc.ba_solve_staged(*la, pyorc.LOCAL_BA_STAGES), reps=5)
q = synth.pose_opt_problem(seed=3000, n=400)
pa = (q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"], q["fx"], q["fy"], q["cx"], q["cy"], q["bf"])
n = len(q["points"]); edges = np.zeros(n, pyorc.EDGE_DTYPE)
edges["pose"] = 0; edges["point"] = np.arange(n); edges["u"] = q["obs"][:, 0]; edges["v"] = q["obs"][:, 1]; edges["ur"] = q["obs"][:, 2]; edges["inv_sigma2"] = q["inv_sigma2"]
cpu_po = lambda: pyorc.ba_solve_staged(q["Tcw0"].reshape(1, 16), np.zeros(1, np.uint8), q["points"], np.ones(n, np.uint8), edges, q["fx"], q["fy"], q["cx"], q["cy"], q["bf"], pyorc.POSE_OPT_STAGES)
row("PoseOptimization (1 frame)", "Optimizer.cc:272-485", "400 observations", lambda: corb.Optimizer.PoseOptimization(*pa), cpu_po)
frames = [(q["Tcw0"], q["points"], q["obs"], q["inv_sigma2"])] * 64
row("PoseOptimization (batch of 64 frames, per call)", "Optimizer.cc:272-485", "64 x 400 observations",
    lambda: corb.Optimizer.PoseOptimizationBatch(frames, q["fx"], q["fy"], q["cx"], q["cy"], q["bf"]), lambda: [cpu_po() for _ in range(64)], reps=5, cpu_reps=1)
s3 = synth.sim3_problem(6000, n=200)
row("OptimizeSim3 (1 candidate)", "Optimizer.cc:1119-1311", "200 correspondences", lambda: corb.Optimizer.OptimizeSim3([s3], 10.0, False), lambda: pyorc.optimize_sim3(s3, 10.0, False))
row("OptimizeSim3 (batch of 16 candidates, per call)", "Optimizer.cc:1119-1311", "16 x 200", lambda: corb.Optimizer.OptimizeSim3([s3] * 16, 10.0, False), lambda: [pyorc.optimize_sim3(s3, 10.0, False) for _ in range(16)], cpu_reps=1)
for K in (100, 300):
    g = synth.essential_graph(7000, K=K)
    row("OptimizeEssentialGraph", "Optimizer.cc:840-1117", "%d keyframes, %d edges" % (K, len(g["vi"])), lambda: corb.Optimizer.OptimizeEssentialGraph(g, 20, False),
        lambda: pyorc.optimize_essential_graph(g, 20, False), reps=3, cpu_reps=1)
rng = np.random.default_rng(1)
sizes = rng.integers(2, 30, 20000); off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32); desc = rng.integers(0, 256, (off[-1], 32), dtype=np.uint8)
row("ComputeDistinctiveDescriptors", "MapPoint.cc:337-402", "20000 map points, %d descriptors" % off[-1], lambda: corb.ComputeDistinctiveDescriptors(desc, off), lambda: pyorc.distinctive_descriptors(desc, off), cpu_reps=1)
T = np.eye(4, dtype=np.float32); T[:3, 3] = [1, 2, 3]; poses = rng.normal(0, 1, (1200, 4, 4)).astype(np.float32); pts = rng.normal(0, 10, (48000, 3)).astype(np.float32)
row("insertServerMapToGlobleMap re-basing", "S/src/MapFusion.cpp:622-658", "1200 keyframes, 48000 map points", lambda: corb.RebaseMap(T, poses, pts), lambda: pyorc.rebase_map(T, poses, pts))
print(json.dumps(dict(note="per-call wall time through the C-ABI incl. transfers; CPU = oracle, 1 thread; MI355X", rows=rows), indent=1))
