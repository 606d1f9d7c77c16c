"""GPU parity tests for the descriptor matchers (bit-exact indices / counts vs the oracle)."""
import numpy as np
import pytest
from test_oracle_match import _make

pytestmark = pytest.mark.gpu


def test_descriptor_distance(corb, pyorc):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (5000, 32), dtype=np.uint8); b = rng.integers(0, 256, (5000, 32), dtype=np.uint8)
    a[0] = 0; b[0] = 255; a[1] = b[1]
    d = corb.ORBmatcher.DescriptorDistance(a, b)
    assert d[0] == 256 and d[1] == 0
    assert np.array_equal(d, np.unpackbits(a ^ b, axis=1).sum(1))
    assert d[7] == pyorc.descriptor_distance(a[7], b[7])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_search_by_bow(corb, pyorc, synth, seed):
    rng = np.random.default_rng(seed)
    for n1, n2, nodes in ((2000, 2000, 100), (300, 500, 7), (64, 64, 1), (1000, 10, 3)):
        d1, a1, v1, fv1, d2, a2, v2, fv2 = _make(rng, synth, n1, n2, nodes)
        for ratio, ori in ((0.75, True), (0.9, False), (0.6, True)):
            m = corb.ORBmatcher(ratio, ori)
            g0, n0 = m.SearchByBoW(dict(desc=d1, angle=a1, valid=v1, fv=fv1), dict(desc=d2, angle=a2, fv=fv2))
            r0, rn0 = pyorc.search_by_bow(0, d1, a1, v1, pyorc.FeatVec(*fv1), d2, a2, v2, pyorc.FeatVec(*fv2), ratio, ori)
            assert np.array_equal(g0, r0) and n0 == rn0
            g1, n1_ = m.SearchByBoW_KFKF(dict(desc=d1, angle=a1, valid=v1, fv=fv1), dict(desc=d2, angle=a2, valid=v2, fv=fv2))
            r1, rn1 = pyorc.search_by_bow(1, d1, a1, v1, pyorc.FeatVec(*fv1), d2, a2, v2, pyorc.FeatVec(*fv2), ratio, ori)
            assert np.array_equal(g1, r1) and n1_ == rn1


def test_search_by_bow_edge_cases(corb, synth):
    rng = np.random.default_rng(3)
    d = synth.correlated_descriptors(10, rng)
    empty_fv = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.uint32))
    m = corb.ORBmatcher(0.7, True)
    g, n = m.SearchByBoW(dict(desc=d, angle=np.zeros(10, np.float32), valid=np.ones(10, np.uint8), fv=empty_fv),
                         dict(desc=d, angle=np.zeros(10, np.float32), fv=empty_fv))
    assert n == 0 and np.all(g == -1)
    fv = (np.array([5], np.uint32), np.array([0, 10], np.int32), np.arange(10, dtype=np.uint32))
    g, n = m.SearchByBoW(dict(desc=d, angle=np.zeros(10, np.float32), valid=np.ones(10, np.uint8), fv=fv),
                         dict(desc=d, angle=np.zeros(10, np.float32), fv=fv))
    assert n == 10 and np.array_equal(g, np.arange(10))      # identical descriptors: each matches itself (ratio vs random 2nd best)


@pytest.mark.parametrize("seed", [4, 5])
def test_search_for_triangulation(corb, pyorc, synth, seed):
    rng = np.random.default_rng(seed)
    n1, n2 = 1500, 1400
    d1 = synth.correlated_descriptors(n1, rng); d2, src = synth.correlated_descriptors(n2, rng, base=d1, flip=0.04)
    kp1 = np.zeros(n1, corb.KP_DTYPE); kp2 = np.zeros(n2, corb.KP_DTYPE)
    kp1["x"], kp1["y"] = rng.uniform(0, 1241, n1), rng.uniform(0, 376, n1)
    kp2["x"] = kp1["x"][src] - rng.uniform(0, 40, n2); kp2["y"] = kp1["y"][src] + rng.normal(0, 0.8, n2)
    kp1["angle"] = rng.uniform(0, 360, n1); kp2["angle"] = (kp1["angle"][src] + rng.normal(0, 15, n2)) % 360
    kp1["octave"] = rng.integers(0, 8, n1); kp2["octave"] = rng.integers(0, 8, n2)
    ur1 = np.where(rng.random(n1) < 0.6, kp1["x"] - 5, -1).astype(np.float32)
    ur2 = np.where(rng.random(n2) < 0.6, kp2["x"] - 5, -1).astype(np.float32)
    mp1 = (rng.random(n1) < 0.3).astype(np.uint8); mp2 = (rng.random(n2) < 0.3).astype(np.uint8)
    fv1 = synth.feature_vector(n1, 20, rng); fv2 = synth.feature_vector(n2, 20, rng)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-4, (3, 3)).astype(np.float32)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32); sigma2 = scale * scale
    for only_stereo in (False, True):
        for ori in (True, False):
            for (ex_, ey_) in ((600.0, 180.0), (-1e6, 188.0)):
                m = corb.ORBmatcher(0.6, ori)
                gp, gn = m.SearchForTriangulation(dict(desc=d1, kp=kp1, u_right=ur1, has_mp=mp1, fv=fv1),
                                                  dict(desc=d2, kp=kp2, u_right=ur2, has_mp=mp2, fv=fv2),
                                                  F12, ex_, ey_, scale, sigma2, only_stereo)
                rp, rn = pyorc.search_for_triangulation(d1, kp1, ur1, mp1, pyorc.FeatVec(*fv1), d2, kp2, ur2, mp2, pyorc.FeatVec(*fv2),
                                                        F12, ex_, ey_, scale, sigma2, only_stereo, ori)
                assert gn == rn and np.array_equal(gp, rp)
