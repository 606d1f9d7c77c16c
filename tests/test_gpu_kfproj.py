"""GPU parity (bit-exact index arrays) of the keyframe-target matchers through the C-ABI vs the oracle:
corb_search_by_projection_reloc, corb_fuse (both overloads), corb_search_by_sim3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher(corb):
    return corb.ORBmatcher(0.6, True)


@pytest.mark.parametrize("seed,n,span", [(5100, 2000, 1.0), (5101, 2000, 0.3), (5102, 700, 1.0), (5103, 3000, 0.5)])
def test_fuse_both_overloads(matcher, pyorc, synth, seed, n, span):
    sc = synth.keyframe_scene(seed, n=n, span=span)
    for th in (3.0, 4.0):
        g = matcher.Fuse(sc["kf2"], sc["T2w"], sc["Ow2"], sc["pts1"], sc["desc1"], th)
        r = pyorc.fuse(sc["kf2"], sc["T2w"], sc["Ow2"], 0, sc["pts1"], sc["desc1"], th)
        assert np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1]) and g[2] == r[2]
        S = sc["T2w"].copy(); S[:3, :] *= np.float32(1.03)                      # a true similarity
        g = matcher.Fuse(sc["kf2"], S, None, sc["pts1"], sc["desc1"], th, sim3=True)
        r = pyorc.fuse(sc["kf2"], S, None, 1, sc["pts1"], sc["desc1"], th)
        assert np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1]) and g[2] == r[2]
    assert r[2] > 50


@pytest.mark.parametrize("seed,n,span", [(5110, 2000, 1.0), (5111, 2000, 0.25), (5112, 500, 1.0)])
def test_reloc_projection(corb, pyorc, synth, seed, n, span):
    sc = synth.keyframe_scene(seed, n=n, span=span)
    for check_ori in (True, False):
        mt = corb.ORBmatcher(0.6, check_ori)
        for th, dist in ((10.0, 100), (3.0, 64)):
            g = mt.SearchByProjection_Reloc(sc["kf2"], sc["claimed2"], sc["T2w"], sc["pts1"], sc["desc1"], th, dist)
            r = pyorc.search_by_projection_reloc(sc["kf2"], sc["claimed2"], sc["T2w"], sc["pts1"], sc["desc1"], th, dist, int(check_ori))
            assert np.array_equal(g[0], r[0]) and g[1] == r[1]
    assert r[1] > 20


@pytest.mark.parametrize("seed,n", [(5120, 2000), (5121, 1200), (5122, 2500)])
def test_search_by_sim3(matcher, pyorc, synth, seed, n):
    sc = synth.keyframe_scene(seed, n=n)
    a = (sc["kf1"], sc["kf2"], sc["T1w"], sc["T2w"], sc["pts1"], sc["desc1"], sc["pts2"], sc["desc2"], sc["s12"], sc["R12"], sc["t12"], 7.5)
    g = matcher.SearchBySim3(*a)
    r = pyorc.search_by_sim3(*a)
    assert np.array_equal(g[0], r[0]) and g[1] == r[1] and r[1] > 100


def test_empty_and_invalid(matcher, pyorc, synth):
    sc = synth.keyframe_scene(5130, n=300)
    pts = sc["pts1"].copy(); pts["valid"] = 0
    g = matcher.Fuse(sc["kf2"], sc["T2w"], sc["Ow2"], pts, sc["desc1"], 3.0)
    assert g[2] == 0 and (g[0] == -1).all()
    g = matcher.Fuse(sc["kf2"], sc["T2w"], sc["Ow2"], pts[:0], sc["desc1"][:0], 3.0)
    assert g[2] == 0 and len(g[0]) == 0
    g = matcher.SearchByProjection_Reloc(sc["kf2"], sc["claimed2"], sc["T2w"], pts, sc["desc1"], 10.0, 100)
    assert g[1] == 0 and (g[0] == -1).all()
