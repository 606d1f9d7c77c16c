"""Development aid: phase times of corb_local_ba_store on a window of the size the configs[2] replay produces (CORB_BA_TIMING=1), next to the host-pointer form."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
import test_gpu_local_ba_store as T
corb.warmup(0)
prob, cm, KF, MP = T._build(corb, synth, 2100, n_local=5, n_fixed=4, ppk=550, outlier_frac=0.03)
K, M = len(cm["kf"]), len(cm["mp_records"])
a = (prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], prob["edges"], prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
print("poses", K, "points", M, "edges", len(prob["edges"]))
N = int(os.environ.get("LBA_PROBE_CALLS", "3"))
def stats(ts): ts = sorted(ts); return "median %.3f  min %.3f ms over %d calls" % (ts[len(ts) // 2], ts[0], len(ts))
hp = []
for i in range(N):
    t0 = time.perf_counter(); g = corb.Optimizer.LocalBundleAdjustment(*a); dt = time.perf_counter() - t0; hp.append(dt * 1e3)
    if i < 3: print("host-pointer call %d: %.2f ms wall" % (i, dt * 1e3), file=sys.stderr, flush=True)
rec = []
for i in range(N):
    t0 = time.perf_counter(); g = corb.LocalBundleAdjustmentStore(KF, np.arange(K), 5, MP, np.arange(M), 1.2, False); dt = time.perf_counter() - t0; rec.append(dt * 1e3)
    if i < 3: print("records call %d: %.2f ms wall, %d erased" % (i, dt * 1e3, len(g["erase"])), file=sys.stderr, flush=True)
if N > 3: print("host-pointer: %s\nrecords:      %s" % (stats(hp[3:]), stats(rec[3:])), file=sys.stderr, flush=True)
