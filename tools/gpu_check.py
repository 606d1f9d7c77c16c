#!/usr/bin/env python3
"""Stage-by-stage parity report HIP vs oracle (run on the GPU box; prints everything, never stops early)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import corbload
corb = corbload.load_pkg()
from corb_slam_amd import synth
from oracle import pyorc

def cmp(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        print("  [%s] SHAPE %s vs %s" % (name, a.shape, b.shape)); return False
    bad = np.argwhere(a != b)
    if len(bad) == 0:
        print("  [%s] ok %s" % (name, a.shape)); return True
    print("  [%s] %d/%d differ; first at %s: gpu=%s ref=%s" % (name, len(bad), a.size, bad[0], a[tuple(bad[0])], b[tuple(bad[0])])); return False

def main():
    print("devices:", corb.device_count())
    L, R = synth.stereo_pair(0)
    t = time.time(); ex = corb.ORBextractor(); print("create %.3fs" % (time.time() - t))
    ref = pyorc.Extractor()
    print("tables:", all(np.array_equal(ex.tables()[k], ref.tables()[k]) for k in ex.tables()))
    t = time.time(); kps, desc = ex(L); print("extract %.3fs n=%d" % (time.time() - t, len(kps)))
    rk, rd = ref.extract(L)
    for l in range(8):
        cmp("pyr%d" % l, ex.pyramid_level(0, l), ref.level(l))
    for l in range(8):
        rb = ref.blurred(l)
        if rb is not None: cmp("blur%d" % l, ex.pyramid_level(0, l, True), rb)
    for l in range(8):
        g = ex.candidates(0, l); r = ref.candidates(l)
        ok = len(g) == len(r) and all(np.array_equal(g[f], r[f]) for f in ("x", "y", "response"))
        print("  [cand%d] %s gpu=%d ref=%d" % (l, "ok" if ok else "DIFF", len(g), len(r)))
        if not ok and len(g) and len(r):
            m = min(len(g), len(r)); d = np.nonzero((g["x"][:m] != r["x"][:m]) | (g["y"][:m] != r["y"][:m]) | (g["response"][:m] != r["response"][:m]))[0]
            if len(d): print("     first diff idx", d[0], g[d[0]], r[d[0]])
    print("  n gpu=%d ref=%d" % (len(kps), len(rk)))
    m = min(len(kps), len(rk))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        cmp("kp." + f, kps[f][:m], rk[f][:m])
    cmp("desc", desc[:m], rd[:m])
    # stereo
    sf = corb.StereoFrontend(max_frames=2)
    for f in range(2):
        l, r = synth.stereo_pair(f); sf.upload(f, l, r)
    sf.run(2); sf.sync()
    for f in range(2):
        out = sf.fetch(f)
        l, r = synth.stereo_pair(f)
        el, er = pyorc.Extractor(), pyorc.Extractor()
        kl, dl = el.extract(l); kr, dr = er.extract(r)
        tb = el.tables()
        ur, dp, n = pyorc.stereo_match(el, er, kl, dl, kr, dr, 386.1448, 718.856, tb["scale"], tb["inv_scale"])
        print(" frame", f, "n_matched gpu=%d ref=%d" % (out["n_matched"], n))
        cmp("f%d.kl" % f, out["kl"].tobytes() == kl.tobytes(), True)
        cmp("f%d.kr" % f, out["kr"].tobytes() == kr.tobytes(), True)
        cmp("f%d.dl" % f, out["dl"], dl); cmp("f%d.dr" % f, out["dr"], dr)
        cmp("f%d.uright" % f, out["u_right"].view(np.uint32), ur.view(np.uint32))
        cmp("f%d.depth" % f, out["depth"].view(np.uint32), dp.view(np.uint32))

if __name__ == "__main__":
    main()
