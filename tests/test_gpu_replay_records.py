"""GPU test: the configs[2] client loop with its tracking stages on device-resident records (tools/replay_client.py, records=True: corb_track_search_last_frame ->
corb_track_pose_optimization(discard) -> corb_track_search_local_points -> corb_track_pose_optimization) -- every stage against the oracle, and the whole run
equal to the host-pointer mode (same matches, same poses, same map)."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed,n_frames", [(9000, 60), (9107, 60)])
def test_records_mode_matches_oracle_and_host_pointer_mode(corb, pyorc, synth, seed, n_frames):
    import replay_client
    reps = {}
    for records in (False, True):
        r = replay_client.Replay(corb, synth, pyorc, n_frames=n_frames, kf_every=3, gba_every=8, seed=seed, images=False, check=True, records=records)
        rep = r.run()
        reps[records] = (rep, r.kfs[-1]["T"].copy(), r.in_map.copy(), r.w.Xest.copy())
        r.close()
        assert rep["errors"] == [], rep["errors"]
        ck = rep["checks_passed"]
        nt = n_frames - 1                                  # every tracked frame's four stages were compared with the oracle
        assert ck["2 SearchByProjection(frame,last)"] == nt and ck["2 PoseOptimization"] == nt and ck["3 SearchByProjection(frame,map)"] == nt and ck["3 PoseOptimization"] == nt
        assert rep["mean"]["matches to the last frame"] > 1000 and rep["final_tracking_error_m"] < 0.05
    a, b = reps[False], reps[True]
    assert a[0]["mean"] == b[0]["mean"] and a[0]["final_tracking_error_m"] == b[0]["final_tracking_error_m"]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
