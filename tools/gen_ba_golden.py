"""Generates tests/golden/ba_config3.json: the oracle's sparse LDL^T global BA (the reference's solver class, Optimizer.cc:43-51 /
GlobalOptimize.cpp:444) on a BASELINE configs[3]-size problem -- 4 clients x 1 200 keyframes (KITTI 00-02 / 04-12 cameras), 100 points per
keyframe, 3..8 observations per point, pixel noise -- non-robust (the server's call) AND with the Huber kernel.  Runs in the BUILD container
(no GPU): the fixture pins the PCG path, which is the only reduced solver used at this size, to an exact factorisation.

    python tools/gen_ba_golden.py            # ~ minutes of one CPU core; writes tests/golden/ba_config3.json

The problem itself is regenerated from its seed by the test (corb_slam_amd.synth.ba_problem_fast is deterministic); a checksum of its arrays is
stored so that a change of the generator cannot go unnoticed."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PROBLEM = dict(n_clients=4, kf_per_client=1200, pts_per_kf=100, seed=4003, obs_range=(3, 8), window=6, pix_noise=1.0)
CAMS = ("00-02", "04-12")
SAMPLE = 64
NAME = "ba_config3"
RUNS = (("nonrobust", False), ("huber", True))
# `python tools/gen_ba_golden.py 12k`: tests/golden/ba_12k.json -- 8 clients x 1 500 keyframes (12 000 keyframes, 1.2 M points, ~6.6 M observations), the server's non-robust call
# only (the forcing-sequence tolerance of the PCG solve pinned to the exact factorisation above 4 800 keyframes); ~1 hour of one CPU core
if len(sys.argv) > 1 and sys.argv[1] == "12k":
    PROBLEM = dict(n_clients=8, kf_per_client=1500, pts_per_kf=100, seed=4012, obs_range=(3, 8), window=6, pix_noise=1.0)
    NAME = "ba_12k"; RUNS = (("nonrobust", False),)


def make_problem(synth):
    kw = dict(PROBLEM); kw["cams"] = [synth.KITTI_CAMS[c] for c in CAMS]
    return synth.ba_problem_fast(**kw)


def checksum(prob):
    h = hashlib.sha256()
    for k in ("poses", "pose_fixed", "points", "point_fixed", "edges", "intr"):
        h.update(np.ascontiguousarray(prob[k]).tobytes())
    return h.hexdigest()


def sample_idx(n, k=SAMPLE):
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))


def main():
    import corbload
    corbload.load_pkg()
    from corb_slam_amd import synth
    from oracle import pyorc
    prob = make_problem(synth)
    args = (prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], prob["edges"], prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
    out = dict(problem=PROBLEM, cams=list(CAMS), checksum=checksum(prob), n_poses=int(len(prob["poses"])), n_points=int(len(prob["points"])), n_edges=int(len(prob["edges"])),
               oracle="oracle/orc_ba.c, solver 2 (block-sparse LDL^T: the reference's LinearSolverEigen class), -O3 -march=native, one thread", runs={})
    pi = sample_idx(len(prob["poses"])); xi = sample_idx(len(prob["points"]))
    out["pose_sample"] = pi.tolist(); out["point_sample"] = xi.tolist()
    path = os.path.join(ROOT, "tests", "golden", NAME + ".json")
    if os.path.exists(path):                       # a run takes ~half an hour: finished runs are kept, an interrupted generation resumes
        old = json.load(open(path))
        if old.get("checksum") == out["checksum"]:
            out["runs"] = old.get("runs", {})
    pyorc.ba_set_solver(2, native=True)
    for tag, robust in RUNS:
        if tag in out["runs"]:
            continue
        t0 = time.perf_counter()
        c = pyorc.ba_solve(*args, iters=10, robust=robust, native=True, intr=prob["intr"])
        dt = time.perf_counter() - t0
        print(tag, "iters", c["iters_done"], "trials", c["trials"], "chi2", c["chi2"][0], "->", c["chi2"][-1], "%.1f s" % dt, flush=True)
        out["runs"][tag] = dict(robust=robust, iters_done=int(c["iters_done"]), trials=int(c["trials"]), wall_s=round(dt, 1),
                                chi2=[float(v) for v in c["chi2"]], lam=[float(v) for v in c["lam"]],
                                poses=np.asarray(c["poses"], np.float64)[pi].reshape(len(pi), 16).tolist(),
                                points=np.asarray(c["points"], np.float64)[xi].tolist())
        json.dump(out, open(path, "w"))
    pyorc.ba_set_solver(0, native=True)
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
