import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
l, r = synth.stereo_pair(0)
ex = corb.ORBextractor(width=1241, height=376, max_images=1)
L = ex.L; h = ex.h
ex(l)
kp = np.zeros(4000, corb.KP_DTYPE); desc = np.zeros((4000, 32), np.uint8); n = C.c_int()
def t(f, reps=20):
    f(); ts=[]
    for _ in range(reps):
        t0=time.perf_counter(); f(); ts.append(time.perf_counter()-t0)
    return 1e3*np.median(ts)
print("upload", t(lambda: L.corb_orb_upload(h, 0, l.ctypes.data_as(C.c_void_p), 1241)))
print("upload+sync", t(lambda: (L.corb_orb_upload(h, 0, l.ctypes.data_as(C.c_void_p), 1241), L.corb_orb_sync(h))))
print("run+sync", t(lambda: (L.corb_orb_run(h, 1), L.corb_orb_sync(h))))
print("fetch", t(lambda: L.corb_orb_fetch(h, 0, kp.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), 4000, C.byref(n))))
print("extract", t(lambda: L.corb_orb_extract(h, l.ctypes.data_as(C.c_void_p), 1241, 376, 1241, kp.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), 4000, C.byref(n))))
print("python ex()", t(lambda: ex(l)))
