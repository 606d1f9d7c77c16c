#!/usr/bin/env python3
"""Generates tests/golden/*.json from the CPU oracle on the deterministic synthetic inputs.

The reference has no golden vectors of its own and cannot be built here, so these
fixtures pin the ORACLE (oracle/*.c) against accidental change; they are data (sizes, counts, SHA-256 of
output arrays, a few leading records), not code.  Re-run only when the oracle's defined semantics change.
"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import corbload
corbload.load_pkg()
from corb_slam_amd import synth
from oracle import pyorc

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

def frame_record(idx, w, h, nfeat):
    L, R = synth.stereo_pair(idx, w, h)
    el, er = pyorc.Extractor(nfeatures=nfeat), pyorc.Extractor(nfeatures=nfeat)
    kl, dl = el.extract(L); kr, dr = er.extract(R)
    tb = el.tables()
    fx, bf = (718.856, 386.1448) if w == 1241 else (1000.0, 500.0)
    ur, dp, nm = pyorc.stereo_match(el, er, kl, dl, kr, dr, bf, fx, tb["scale"], tb["inv_scale"])
    return dict(frame=idx, width=w, height=h, nfeatures=nfeat, fx=fx, bf=bf,
                image_sha=dict(left=sha(L), right=sha(R)),
                n_left=int(len(kl)), n_right=int(len(kr)), n_matched=int(nm),
                per_level_left=[int(el.level_count(l)) for l in range(8)],
                candidates_left=[int(len(el.candidates(l))) for l in range(8)],
                sha=dict(kp_left=sha(kl), desc_left=sha(dl), kp_right=sha(kr), desc_right=sha(dr), u_right=sha(ur), depth=sha(dp),
                         pyr7_left=sha(el.level(7)), blur0_left=sha(el.blurred(0))),
                head_left=[[float(k["x"]), float(k["y"]), float(k["angle"]), float(k["response"]), int(k["octave"])] for k in kl[:4]],
                head_desc_left=dl[0].tolist())

def main():
    os.makedirs(OUT, exist_ok=True)
    recs = [frame_record(0, 1241, 376, 2000), frame_record(1, 1241, 376, 2000), frame_record(7, 1241, 376, 2000)]
    json.dump(dict(generator="tools/gen_golden.py", note="oracle outputs on corb-slam_amd/synth.py stereo_pair()", frames=recs),
              open(os.path.join(OUT, "orb_stereo_kitti.json"), "w"), indent=1)
    big = [frame_record(3, 1920, 1080, 4000)]
    json.dump(dict(generator="tools/gen_golden.py", frames=big), open(os.path.join(OUT, "orb_stereo_1080p.json"), "w"), indent=1)
    # tables pinned by the reference's own arithmetic
    ex = pyorc.Extractor(); ex.extract(synth.stereo_pair(0)[0]); tb = ex.tables()
    json.dump(dict(quota=tb["quota"].tolist(), umax=tb["umax"].tolist(),
                   scale_bits=[int(v) for v in tb["scale"].view(np.uint32)],
                   level_dims=[list(ex.level(l).shape[::-1]) for l in range(8)]),
              open(os.path.join(OUT, "tables_kitti.json"), "w"), indent=1)
    print("golden written to", OUT)

if __name__ == "__main__":
    main()
