"""Single-call latency of the drop-in operators (what a client sees before it batches): ORBextractor::operator() on one image and one
stereo frame through StereoFrontend, host buffers in and out."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
l, r = synth.stereo_pair(0)
ex = corb.ORBextractor(width=1241, height=376, max_images=1)
ex(l)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); k, d = ex(l); ts.append(time.perf_counter() - t0)
print("ORBextractor::operator() 1 image: median %.3f ms (%d keypoints)" % (1e3 * np.median(ts), len(k)))
sf = corb.StereoFrontend(nfeatures=2000, width=1241, height=376, max_frames=1, fx=718.856, bf=386.1448)
sf.upload(0, l, r); sf.run(1); sf.sync(); sf.fetch(0)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); sf.upload(0, l, r); sf.run(1); sf.sync(); o = sf.fetch(0); ts.append(time.perf_counter() - t0)
print("stereo frame (upload + extract L/R + ComputeStereoMatches + fetch): median %.3f ms (%d matches)" % (1e3 * np.median(ts), o["n_matched"]))
ts = []
for _ in range(30):
    t0 = time.perf_counter(); sf.run(1); sf.sync(); ts.append(time.perf_counter() - t0)
print("stereo frame device part only (run + sync): median %.3f ms" % (1e3 * np.median(ts)))
