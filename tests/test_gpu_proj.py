"""GPU parity: projection-guided matchers vs the oracle -- identical match arrays and counts (the greedy, order-dependent
assignment of the reference is reproduced exactly by the round-based resolver)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n,dense", [(4001, 700, False), (4005, 2000, False), (4006, 2000, True), (4007, 64, False)])
def test_search_by_projection_map(corb, pyorc, synth, seed, n, dense):
    s = synth.tracking_scene(seed=seed, n=n, dense=dense)
    for th, ratio in ((1.0, 0.8), (3.0, 0.8), (5.0, 0.9)):
        g, gn = corb.ORBmatcher(ratio, True).SearchByProjection(s["cur"], s["mps"], s["last_desc"], th)
        r, rn = pyorc.search_by_projection_map(s["cur"], s["mps"], s["last_desc"], th, ratio)
        assert gn == rn and np.array_equal(g, r)
    assert rn > 0


@pytest.mark.parametrize("seed,motion,dense", [(4002, (0, 0, 0.8), False), (4003, (0, 0, -0.9), False), (4004, (0.3, 0, 0.1), False), (4008, (0, 0, 0.8), True)])
def test_search_by_projection_frame(corb, pyorc, synth, seed, motion, dense):
    s = synth.tracking_scene(seed=seed, n=2000, motion=motion, dense=dense)
    for mono, th, ori in ((0, 7.0, True), (0, 15.0, True), (1, 15.0, True), (0, 7.0, False)):
        g, gn = corb.ORBmatcher(0.9, ori).SearchByProjection_Frame(s["cur"], s["Tcw"], s["Tlw"], s["fx"], s["fy"], s["cx"], s["cy"], s["bf"], s["mb"],
                                                                   s["last"], s["last_desc"], th, mono)
        r, rn = pyorc.search_by_projection_frame(s["cur"], s["Tcw"], s["Tlw"], s["fx"], s["fy"], s["cx"], s["cy"], s["bf"], s["mb"],
                                                 s["last"], s["last_desc"], th, mono, int(ori))
        assert gn == rn and np.array_equal(g, r)
    assert rn > 100


def test_projection_edge_cases(corb, pyorc, synth):
    s = synth.tracking_scene(seed=4009, n=300)
    none = s["mps"].copy(); none["valid"] = 0
    g, gn = corb.ORBmatcher(0.8, True).SearchByProjection(s["cur"], none, s["last_desc"], 3.0)
    assert gn == 0 and np.all(g == -1)
    allc = dict(s["cur"]); allc["claimed"] = np.ones_like(s["cur"]["claimed"])
    g, gn = corb.ORBmatcher(0.8, True).SearchByProjection(allc, s["mps"], s["last_desc"], 3.0)
    assert gn == 0
    # every query projects onto the same spot: one long claim chain (worst case for the round-based resolver)
    chain = s["mps"].copy(); chain["valid"] = 1; chain["claims"] = 1; chain["proj_x"] = 600; chain["proj_y"] = 180; chain["proj_xr"] = 590; chain["level"] = 3
    g, gn = corb.ORBmatcher(0.95, True).SearchByProjection(s["cur"], chain, s["last_desc"], 8.0)
    r, rn = pyorc.search_by_projection_map(s["cur"], chain, s["last_desc"], 8.0, 0.95)
    assert gn == rn and np.array_equal(g, r)
