"""GPU parity (bit-exact index arrays) of the keyframe-target matchers through the C-ABI vs the oracle:
corb_search_by_projection_reloc, corb_search_by_projection_scw, corb_fuse (both overloads), corb_search_by_sim3."""
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


@pytest.mark.parametrize("seed,n,span", [(5140, 2000, 1.0), (5141, 2000, 0.25), (5142, 500, 1.0), (5143, 3000, 0.4)])
def test_search_by_projection_scw(matcher, pyorc, synth, seed, n, span):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:425-538): the loop / fusion event's projection matcher; span < 1 crowds the
    points so that the sequential claims of keyframe features (:510, :530) decide most matches"""
    sc = synth.keyframe_scene(seed, n=n, span=span)
    if seed & 1:
        sc = synth.crowd_keyframe_scene(sc, seed)                               # repeated texture: several features within TH_LOW of a point, points compete
    total = 0
    for th, s in ((10.0, 1.0), (10.0, 1.04), (4.0, 0.97)):
        S = sc["T2w"].copy(); S[:3, :] *= np.float32(s)                         # Scw = [s R | s t]
        for claimed in (sc["claimed2"], np.zeros(n, np.uint8)):
            g = matcher.SearchByProjection_Scw(sc["kf2"], claimed, S, sc["pts1"], sc["desc1"], th)
            r = pyorc.search_by_projection_scw(sc["kf2"], claimed, S, sc["pts1"], sc["desc1"], th)
            assert np.array_equal(g[0], r[0]) and g[1] == r[1]
            assert (g[0][claimed != 0] == -1).all() and g[1] == (g[0] >= 0).sum()
            m = g[0][g[0] >= 0]; assert len(np.unique(m)) == len(m)               # a point is written into one feature at most
            total += g[1]
    assert total > 100


def test_search_by_projection_scw_differs_from_the_unclaimed_minimum(matcher, pyorc, synth):
    """the order dependence is real on this input: some points lose their best feature to an earlier point and take another one (or none) -- the independent
    per-point minimum (Fuse's search with the same gates) names a different feature for them"""
    sc = synth.crowd_keyframe_scene(synth.keyframe_scene(5145, n=3000, span=0.3), 5145)
    S = sc["T2w"].copy()
    g = matcher.SearchByProjection_Scw(sc["kf2"], None, S, sc["pts1"], sc["desc1"], 10.0)
    r = pyorc.search_by_projection_scw(sc["kf2"], np.zeros(3000, np.uint8), S, sc["pts1"], sc["desc1"], 10.0)
    assert np.array_equal(g[0], r[0]) and g[1] == r[1] and g[1] > 200
    bi, bd, nf = matcher.Fuse(sc["kf2"], S, None, sc["pts1"], sc["desc1"], 10.0, sim3=True)
    feat_of_point = np.full(3000, -1); feat_of_point[g[0][g[0] >= 0]] = np.nonzero(g[0] >= 0)[0]
    assert ((feat_of_point >= 0) & (bi >= 0) & (feat_of_point != bi)).sum() > 0


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
    # SearchByProjection(KeyFrame*, Scw, ...): every point already found / bad, no points at all, every feature already matched
    g = matcher.SearchByProjection_Scw(sc["kf2"], sc["claimed2"], sc["T2w"], pts, sc["desc1"], 10.0)
    assert g[1] == 0 and (g[0] == -1).all()
    g = matcher.SearchByProjection_Scw(sc["kf2"], sc["claimed2"], sc["T2w"], pts[:0], sc["desc1"][:0], 10.0)
    assert g[1] == 0 and (g[0] == -1).all() and len(g[0]) == 300
    g = matcher.SearchByProjection_Scw(sc["kf2"], np.ones(300, np.uint8), sc["T2w"], sc["pts1"], sc["desc1"], 10.0)
    r = pyorc.search_by_projection_scw(sc["kf2"], np.ones(300, np.uint8), sc["T2w"], sc["pts1"], sc["desc1"], 10.0)
    assert g[1] == 0 and r[1] == 0 and (g[0] == -1).all()


def _init_frames(synth, seed, n, span=1.0, crowd=False):
    return synth.monocular_init_pair(seed, n=n, span=span, crowd=crowd)


@pytest.mark.parametrize("seed,n,span,crowd", [(7100, 1500, 1.0, False), (7101, 2000, 0.5, True), (7102, 800, 0.3, True), (7103, 3000, 1.0, True)])
def test_search_for_initialization(corb, pyorc, synth, seed, n, span, crowd):
    """SearchForInitialization (ORBmatcher.cc:540-655) against the oracle: vnMatches12, vbPrevMatched and the count, for both orientation settings and two ratios; the crowded
    scenes (repeated texture) make later features take matches away from earlier ones (:583, :601-605); a second call starts from the first call's vbPrevMatched"""
    f1, f2, pm, src = _init_frames(synth, seed, n, span, crowd)
    total = 0
    for check_ori in (True, False):
        for ratio, win in ((0.9, 100), (0.7, 40)):
            mt = corb.ORBmatcher(ratio, check_ori)
            g = mt.SearchForInitialization(f1, f2, pm, win)
            r = pyorc.search_for_initialization(f1, f2, pm, win, ratio, check_ori)
            assert np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1]) and g[2] == r[2] == int((g[0] >= 0).sum())
            assert (g[0][f1["keys_un"]["octave"] > 0] == -1).all()
            m = g[0][g[0] >= 0]; assert len(np.unique(m)) == len(m)
            g2 = mt.SearchForInitialization(f1, f2, g[1], win)
            r2 = pyorc.search_for_initialization(f1, f2, r[1], win, ratio, check_ori)
            assert np.array_equal(g2[0], r2[0]) and np.array_equal(g2[1], r2[1]) and g2[2] == r2[2]
            total += g[2]
    assert total > 200


def test_search_for_initialization_edge_cases(corb, pyorc, synth):
    f1, f2, pm, _ = _init_frames(synth, 7110, 400)
    mt = corb.ORBmatcher(0.9, True)
    e1 = dict(f1); e1["keys_un"] = f1["keys_un"][:0]; e1["desc"] = f1["desc"][:0]
    g = mt.SearchForInitialization(e1, f2, pm[:0], 100)
    assert g[2] == 0 and len(g[0]) == 0
    k = f1["keys_un"].copy(); k["octave"] = 2; h1 = dict(f1); h1["keys_un"] = k              # no level-0 feature
    g = mt.SearchForInitialization(h1, f2, pm, 100)
    assert g[2] == 0 and (g[0] == -1).all() and np.array_equal(g[1], pm)
    g = mt.SearchForInitialization(f1, f2, pm + np.float32(5000.0), 100)                          # every window outside the image
    r = pyorc.search_for_initialization(f1, f2, pm + np.float32(5000.0), 100, 0.9, True)
    assert g[2] == 0 == r[2] and (g[0] == -1).all()
    g = mt.SearchForInitialization(f1, f2, pm, 0)                                                 # an empty window
    r = pyorc.search_for_initialization(f1, f2, pm, 0, 0.9, True)
    assert np.array_equal(g[0], r[0]) and g[2] == r[2]


class _Product:
    """the oracle module's call names on the product's matcher objects, so tools/gen_matcher_golden.py's case list runs on either"""
    def __init__(self, corb):
        self.corb = corb; self.m = corb.ORBmatcher(0.6, True)

    def fuse(self, kf, T, Ow, sim3, pts, desc, th):
        return self.m.Fuse(kf, T, Ow, pts, desc, th, sim3=bool(sim3)) if sim3 else self.m.Fuse(kf, T, Ow, pts, desc, th)

    def search_by_projection_reloc(self, kf, claimed, T, pts, desc, th, dist, check_ori):
        return self.corb.ORBmatcher(0.6, bool(check_ori)).SearchByProjection_Reloc(kf, claimed, T, pts, desc, th, dist)

    def search_by_sim3(self, *a):
        return self.m.SearchBySim3(*a)

    def search_by_projection_scw(self, kf, claimed, S, pts, desc, th):
        return self.m.SearchByProjection_Scw(kf, claimed, S, pts, desc, th)

    def search_for_initialization(self, f1, f2, pm, win, ratio, check_ori):
        return self.corb.ORBmatcher(ratio, bool(check_ori)).SearchForInitialization(f1, f2, pm, win)


def test_matchers_match_committed_golden(corb, synth):
    """the product against tests/golden/matchers.json (committed oracle outputs, tools/gen_matcher_golden.py) -- no live oracle in this comparison"""
    import importlib.util, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_matcher_golden", os.path.join(root, "tools", "gen_matcher_golden.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    rec = json.load(open(os.path.join(root, "tests", "golden", "matchers.json")))
    prod = _Product(corb)
    for name, params, fn in g.cases(synth):
        assert g.digest(fn(prod)) == rec["cases"][name]["out"], name
