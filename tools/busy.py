"""Development aid: `import busy; busy.start(corb, synth)` keeps the GPU busy with stereo front-end runs from a background thread (CORB_TEST_BUSY=1 only), so that the
stress tools' kernels share the compute units with another stream's"""
import os, threading
import numpy as np
_state = {}
def start(corb, synth, n=16):
    if not os.environ.get("CORB_TEST_BUSY") or _state: return
    fr = [synth.stereo_pair(900 + i, w=1241, h=376) for i in range(4)]
    P = np.ascontiguousarray(np.stack([np.stack(fr[i % 4]) for i in range(n)]))
    sf = corb.StereoFrontend(nfeatures=2000, width=1241, height=376, max_frames=n); sf.upload_batch(0, P)
    def bg():
        while True: sf.run(n); sf.sync()
    t = threading.Thread(target=bg, daemon=True); t.start(); _state["t"] = t
