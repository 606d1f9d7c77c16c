"""GPU box: extract + stereo-match throughput at the image size of BASELINE config 5 (1920x1080, 4000 features; fx = 1000, bf = 500 chosen here)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sf = corb.StereoFrontend(nfeatures=4000, width=1920, height=1080, max_frames=B, fx=1000.0, bf=500.0)
frames = [synth.stereo_pair(i, 1920, 1080) for i in range(8)]
for s in range(B):
    l, r = frames[s % 8]; sf.upload(s, l, r)
sf.sync()
for _ in range(3): sf.run(B)
sf.sync()
t0 = time.perf_counter(); N = 20
for _ in range(N): sf.run(B)
sf.sync()
dt = (time.perf_counter() - t0) / N
o = sf.fetch(0)
print("1920x1080 / 4000 features: %d frames per step, %.3f ms per step -> %.0f stereo frames/s (%d / %d keypoints, %d stereo matches in frame 0)" % (
    B, dt * 1e3, B / dt, len(o["kl"]), len(o["kr"]), o["n_matched"]))
