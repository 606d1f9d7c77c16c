"""Development aid: the two matchers added in round 6 against the oracle on many random scenes (bit-exact index arrays), beyond the seeds the tests hold"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
from oracle import pyorc
pyorc.build()
import busy; busy.start(corb, synth)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(99)
bad = 0; tot = 0
mt = corb.ORBmatcher(0.6, True)
for it in range(N):
    seed = 20000 + it; n = int(rng.integers(300, 3000)); span = float(rng.choice([1.0, 0.5, 0.3, 0.2]))
    sc = synth.keyframe_scene(seed, n=n, span=span)
    if it & 1: sc = synth.crowd_keyframe_scene(sc, seed, frac=float(rng.choice([0.3, 0.6])))
    S = sc["T2w"].copy(); S[:3, :] *= np.float32(rng.choice([1.0, 1.05, 0.93]))
    th = float(rng.choice([10.0, 4.0, 15.0]))
    claimed = sc["claimed2"] if it % 3 else np.zeros(n, np.uint8)
    g = mt.SearchByProjection_Scw(sc["kf2"], claimed, S, sc["pts1"], sc["desc1"], th)
    r = pyorc.search_by_projection_scw(sc["kf2"], claimed, S, sc["pts1"], sc["desc1"], th)
    tot += 1
    if not (np.array_equal(g[0], r[0]) and g[1] == r[1]): bad += 1; print("scw mismatch", seed, n, span, th, g[1], r[1])
    f1, f2, pm, _ = synth.monocular_init_pair(seed, n=n, span=span, crowd=bool(it & 1), steal_frac=float(rng.choice([0.0, 0.15, 0.4])))
    ratio = float(rng.choice([0.9, 0.7])); win = int(rng.choice([100, 40, 160])); ori = bool(it & 2)
    g = corb.ORBmatcher(ratio, ori).SearchForInitialization(f1, f2, pm, win)
    r = pyorc.search_for_initialization(f1, f2, pm, win, ratio, ori)
    tot += 1
    if not (np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1]) and g[2] == r[2]): bad += 1; print("init mismatch", seed, n, span, ratio, win, ori, g[2], r[2])
print("cases", tot, "mismatches", bad)
