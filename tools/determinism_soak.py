"""GPU box: the same batch through the extract+match path many times; every run must produce the same bytes (a data race in a
kernel shows up as a run-to-run difference)."""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(7)
sf = corb.StereoFrontend(max_frames=B)
for s in range(B):
    if s % 8 == 7:      # some maximum-density frames: the quadtree's global-key path
        l = rng.integers(0, 256, (376, 1241), dtype=np.uint8); r = np.roll(l, -7, axis=1)
    else:
        l, r = synth.stereo_pair(s)
    sf.upload(s, l, r)
sf.sync()
ref = None; bad = 0
for it in range(reps):
    sf.run(B); sf.sync()
    o = sf.fetch_batch(0, B)
    hsh = hashlib.sha256()
    n = o["counts"]
    hsh.update(n.tobytes())
    for i in range(2 * B):
        hsh.update(o["kp"][i, : n[i]].tobytes()); hsh.update(o["desc"][i, : n[i]].tobytes())
    for f in range(B):
        hsh.update(o["u_right"][f, : n[2 * f]].tobytes()); hsh.update(o["depth"][f, : n[2 * f]].tobytes())
    d = hsh.hexdigest()
    if ref is None: ref = d
    elif d != ref: bad += 1
print("runs %d batch %d mismatching %d digest %s" % (reps, B, bad, ref[:16]))
sys.exit(1 if bad else 0)
