"""GPU test: BASELINE configs[2] -- one client's Tracking + LocalMapping loop and the global BA on one GPU, every stage checked against the oracle run of the
same inputs (tools/replay_client.py: stereo front-end -> SearchByProjection -> PoseOptimization -> keyframe -> SearchForTriangulation -> Fuse ->
LocalBundleAdjustment -> GlobalBundleAdjustemnt)."""
import os
import sys
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_client_loop_matches_oracle_stage_by_stage(corb, pyorc, synth):
    import replay_client
    r = replay_client.Replay(corb, synth, pyorc, n_frames=26, kf_every=3, gba_every=4, images=True, check=True)
    rep = r.run()
    r.close()
    assert rep["errors"] == [], rep["errors"]
    ck = rep["checks_passed"]
    # every stage ran and was compared: 25 tracked frames, 9 keyframes (frames 0, 3, ..., 24), two global adjustments (after 4 and 8 keyframes)
    assert ck["2 SearchByProjection(frame,last)"] == 25 and ck["2 PoseOptimization"] == 25 and ck["3 SearchByProjection(frame,map)"] == 25 and ck["3 PoseOptimization"] == 25
    assert ck["1 stereo front-end"] >= 4 and ck["4 SearchForTriangulation"] >= 15 and ck["4 SearchForTriangulation (store slots)"] == ck["4 SearchForTriangulation"]
    assert ck["5 Fuse"] == 8 and ck["6 LocalBundleAdjustment"] == 7 and ck["7 GlobalBundleAdjustemnt"] == 2
    assert rep["keyframes"] == 9 and rep["map_points"] > 3000
    m = rep["mean"]
    assert m["matches to the last frame"] > 1000 and m["triangulation pairs with the right landmark"] > 10 and m["fused points"] > 10
    assert rep["final_tracking_error_m"] < 0.05          # the loop tracks the synthetic trajectory (3 cm landmark noise)
