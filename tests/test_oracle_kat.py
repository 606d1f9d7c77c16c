"""CPU tests: the oracle against known answers, the tables the reference's own arithmetic pins
(quotas, umax, pyramid sizes), independent restatements, and the committed goldens."""
import hashlib
import json
import math
import os
import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_hamming_kats(pyorc):
    z = np.zeros(32, np.uint8); o = np.full(32, 255, np.uint8)
    assert pyorc.descriptor_distance(z, z) == 0
    assert pyorc.descriptor_distance(z, o) == 256
    for bit in (0, 7, 8, 100, 255):
        b = z.copy(); b[bit // 8] = 1 << (bit % 8)
        assert pyorc.descriptor_distance(z, b) == 1
    rng = np.random.default_rng(5)
    for _ in range(50):
        a, b = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
        assert pyorc.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())


def test_tables_match_reference_arithmetic(pyorc):
    """Values derivable from ORBextractor.cc:415-469, 1111-1113 alone."""
    ex = pyorc.Extractor(2000, 1.2, 8, 20, 7)
    tb = ex.tables()
    assert tb["quota"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert tb["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    s = np.float32(1.0)
    for i in range(8):
        assert tb["scale"][i] == s
        assert tb["inv_scale"][i] == np.float32(1.0) / s
        assert tb["sigma2"][i] == s * s
        s = s * np.float32(1.2)
    ex.extract(np.zeros((376, 1241), np.uint8))
    dims = [ex.level(l).shape[::-1] for l in range(8)]
    assert dims == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    ex4 = pyorc.Extractor(4000, 1.2, 8, 20, 7)
    assert ex4.tables()["quota"].tolist() == [869, 724, 603, 503, 419, 349, 291, 242]
    ex4.extract(np.zeros((1080, 1920), np.uint8))
    assert [ex4.level(l).shape[::-1] for l in range(8)] == [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)]
    g = json.load(open(os.path.join(GOLD, "tables_kitti.json")))
    assert g["quota"] == tb["quota"].tolist() and g["umax"] == tb["umax"].tolist()
    assert g["scale_bits"] == [int(v) for v in tb["scale"].view(np.uint32)]


def test_gaussian_taps_and_blur(pyorc):
    img = np.zeros((40, 40), np.uint8); img[20, 20] = 255
    b = pyorc.gaussian_blur7(img).astype(np.int64)
    K = np.array([18, 34, 49, 55, 49, 34, 18])
    expect = ((np.outer(K, K) * 255 + (1 << 15)) >> 16)
    assert np.array_equal(b[17:24, 17:24], expect)
    c = np.full((30, 50), 77, np.uint8)
    out = pyorc.gaussian_blur7(c)             # taps sum to 257/256 per axis: constant 77 -> (77*257*257+2^15)>>16
    assert np.all(out == ((77 * 257 * 257 + (1 << 15)) >> 16))
    # reflect-101 border: independent numpy restatement
    rng = np.random.default_rng(3); im = rng.integers(0, 256, (23, 31), dtype=np.uint8)
    pad = np.pad(im.astype(np.int64), 3, mode="reflect")
    row = sum(K[k] * pad[:, k:k + 31] for k in range(7))
    col = sum(K[k] * row[k:k + 23, :] for k in range(7))
    assert np.array_equal(pyorc.gaussian_blur7(im), np.clip((col + (1 << 15)) >> 16, 0, 255).astype(np.uint8))


def test_resize_properties(pyorc):
    c = np.full((100, 120), 200, np.uint8)
    assert np.all(pyorc.resize_linear(c, 100, 83) == 200)
    rng = np.random.default_rng(1); im = rng.integers(0, 256, (60, 72), dtype=np.uint8)
    assert np.array_equal(pyorc.resize_linear(im, 72, 60), im)       # identity scale
    # independent restatement (numpy, same fixed-point definition) on a 1.2x shrink
    sw, sh, dw, dh = 72, 60, 60, 50
    def coefs(d, s):
        sc = 1.0 / (d / s); out = []
        for i in range(d):
            f = np.float32((i + 0.5) * sc - 0.5); si = int(math.floor(f)); f = np.float32(f - np.float32(si))
            if si < 0: si, f = 0, np.float32(0)
            if si >= s - 1: si, f = s - 1, np.float32(0)
            out.append((si, int(np.rint((np.float32(1.0) - f) * np.float32(2048))), int(np.rint(f * np.float32(2048)))))
        return out
    cx, cy = coefs(dw, sw), coefs(dh, sh)
    ref = np.zeros((dh, dw), np.uint8); S = im.astype(np.int64)
    for y, (sy, b0, b1) in enumerate(cy):
        sy1 = min(sy + 1, sh - 1)
        for x, (sx, a0, a1) in enumerate(cx):
            sx1 = min(sx + 1, sw - 1)
            d0 = S[sy, sx] * a0 + S[sy, sx1] * a1; d1 = S[sy1, sx] * a0 + S[sy1, sx1] * a1
            ref[y, x] = ((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2) & 255
    assert np.array_equal(pyorc.resize_linear(im, dw, dh), ref)


def _fast_bruteforce(img, t):
    """FAST-9/16 straight from the definition: corner test, score = max t' still a corner, strict 8-NMS."""
    h, w = img.shape
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]; dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    I = img.astype(np.int64); score = np.zeros((h, w), np.int64)
    def is_corner(y, x, th):
        v = I[y, x]; ring = [I[y + dy[k], x + dx[k]] for k in range(16)]
        for sign in (1, -1):
            f = [(sign * (r - v)) > th for r in ring]; run = 0
            for k in range(32):
                run = run + 1 if f[k % 16] else 0
                if run >= 9: return True
        return False
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if is_corner(y, x, t):
                s = t
                while s < 255 and is_corner(y, x, s + 1): s += 1
                score[y, x] = s
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s > 0 or (s == 0 and False):
                nb = score[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                if is_corner(y, x, t) and np.all(s > nb): out.append((x, y, s))
    return out


def test_fast_against_definition(pyorc):
    rng = np.random.default_rng(11)
    img = rng.integers(90, 140, (40, 44)).astype(np.uint8)
    img[10:20, 12:25] = 220; img[25:33, 5:15] = 20; img[8, 30] = 255
    for t in (7, 20):
        got = pyorc.fast(img, t, True)
        exp = _fast_bruteforce(img, t)
        assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == exp
        assert len(exp) > 0


def test_fast_atan2_and_sincos(pyorc):
    rng = np.random.default_rng(2)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1000
        a = pyorc.fast_atan2(y, x); r = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - r); d = min(d, 360 - d)
        assert d < 0.02, (y, x, a, r)            # polynomial accuracy ~0.0035 deg
    assert pyorc.fast_atan2(0.0, 0.0) == 0.0
    assert pyorc.fast_atan2(0.0, 1.0) == 0.0 and abs(pyorc.fast_atan2(1.0, 0.0) - 90.0) < 1e-3
    xs = rng.uniform(0, 2 * math.pi, 20000).astype(np.float32)
    ulp_bad = 0
    for x in xs:
        s, c = pyorc.sincosf(x)
        if np.float32(math.sin(float(x))) != np.float32(s): ulp_bad += 1
        if np.float32(math.cos(float(x))) != np.float32(c): ulp_bad += 1
    # defined as the correctly-rounded value of the double-precision result; vs libm double -> float: identical
    assert ulp_bad == 0


def test_ic_angle_symmetry(pyorc):
    img = np.zeros((64, 64), np.uint8); img[:, 32:] = 200          # bright on +x => angle ~0
    assert abs(pyorc.ic_angle(img, 32, 32)) < 1.0 or abs(pyorc.ic_angle(img, 32, 32) - 360) < 1.0
    img = np.zeros((64, 64), np.uint8); img[32:, :] = 200          # bright on +y => ~90 deg
    assert abs(pyorc.ic_angle(img, 32, 32) - 90.0) < 1.0


def test_octree_tie_break_is_creation_order(pyorc):
    """Two equal-size candidate nodes: the defined order expands the later-created one first
    (the reference orders by heap address there, ORBextractor.cc:684)."""
    kps = np.zeros(12, pyorc.KP_DTYPE)
    # root 0..200 x 0..100 -> nIni=2 ; put 3 points in each of 4 quadrant-ish clusters
    pts = [(10, 10), (12, 14), (14, 30), (60, 10), (62, 12), (64, 30), (110, 10), (112, 12), (114, 40), (160, 60), (162, 62), (164, 90)]
    for i, (x, y) in enumerate(pts):
        kps[i]["x"], kps[i]["y"], kps[i]["response"] = x, y, 10 + i
    out5 = pyorc.distribute_octree(kps, 16, 216, 16, 116, 5)
    out4 = pyorc.distribute_octree(kps, 16, 216, 16, 116, 4)
    assert len(out4) >= 4 and len(out5) >= 5
    # determinism + every output is an input
    again = pyorc.distribute_octree(kps, 16, 216, 16, 116, 5)
    assert again.tobytes() == out5.tobytes()
    assert set(map(tuple, np.stack([out5["x"], out5["y"]], 1).tolist())) <= set(map(tuple, [(float(a), float(b)) for a, b in pts]))


def test_octree_array_formulation_equals_serial(pyorc):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("octree_proto", os.path.join(root, "tools", "octree_proto.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert m._selftest(trials=120, seed=9)


def test_extract_edge_cases(pyorc, synth):
    ex = pyorc.Extractor()
    k, d = ex.extract(np.zeros((0, 0), np.uint8))
    assert len(k) == 0
    k, d = ex.extract(synth.flat_image(1241, 376))
    assert len(k) == 0 and d.shape == (0, 32)
    rng = np.random.default_rng(0)
    k, d = ex.extract(rng.integers(0, 256, (376, 1241), dtype=np.uint8))   # dense corners everywhere
    assert 2000 <= len(k) <= 2000 + 3 * 8
    assert np.all(k["octave"][:-1] <= k["octave"][1:])                      # levels concatenated in order
    assert np.all(k["class_id"] == -1) and np.all((k["angle"] >= 0) & (k["angle"] < 360.0001))


@pytest.mark.parametrize("name", ["orb_stereo_kitti.json", "orb_stereo_1080p.json"])
def test_oracle_matches_golden(pyorc, synth, name):
    g = json.load(open(os.path.join(GOLD, name)))
    for rec in g["frames"][: (2 if "kitti" in name else 1)]:
        L, R = synth.stereo_pair(rec["frame"], rec["width"], rec["height"])
        assert sha(L) == rec["image_sha"]["left"] and sha(R) == rec["image_sha"]["right"]
        el, er = pyorc.Extractor(nfeatures=rec["nfeatures"]), pyorc.Extractor(nfeatures=rec["nfeatures"])
        kl, dl = el.extract(L); kr, dr = er.extract(R)
        tb = el.tables()
        ur, dp, nm = pyorc.stereo_match(el, er, kl, dl, kr, dr, rec["bf"], rec["fx"], tb["scale"], tb["inv_scale"])
        assert (len(kl), len(kr), nm) == (rec["n_left"], rec["n_right"], rec["n_matched"])
        assert [el.level_count(l) for l in range(8)] == rec["per_level_left"]
        assert sha(kl) == rec["sha"]["kp_left"] and sha(dl) == rec["sha"]["desc_left"]
        assert sha(kr) == rec["sha"]["kp_right"] and sha(dr) == rec["sha"]["desc_right"]
        assert sha(ur) == rec["sha"]["u_right"] and sha(dp) == rec["sha"]["depth"]


def test_stereo_disparities_are_the_synthetic_ones(pyorc, synth):
    """Property: matched disparities equal the generator's per-band integer shifts (+- sub-pixel)."""
    L, R = synth.stereo_pair(2)
    el, er = pyorc.Extractor(), pyorc.Extractor()
    kl, dl = el.extract(L); kr, dr = er.extract(R)
    tb = el.tables()
    ur, dp, nm = pyorc.stereo_match(el, er, kl, dl, kr, dr, 386.1448, 718.856, tb["scale"], tb["inv_scale"])
    assert nm > 200
    disp = (kl["x"] - ur)[ur >= 0]
    assert np.all(disp > 0) and np.all(disp < 718.9)
    near_int = np.abs(disp - np.rint(disp)) < 0.6 * tb["scale"][kl["octave"][ur >= 0]] + 0.6
    assert near_int.mean() > 0.9
    assert np.allclose(dp[ur >= 0], np.float32(386.1448) / disp, rtol=1e-6)


def _matcher_golden_mod():
    import importlib.util, os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_matcher_golden.py")
    spec = importlib.util.spec_from_file_location("gen_matcher_golden", p)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_oracle_matchers_match_golden(pyorc, synth):
    """tests/golden/matchers.json (tools/gen_matcher_golden.py): the keyframe-target matchers' outputs on fixed synthetic scenes -- the oracle has not drifted"""
    import json, os
    g = _matcher_golden_mod()
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "matchers.json")))
    names = []
    for name, params, fn in g.cases(synth):
        assert rec["cases"][name]["params"] == json.loads(json.dumps(params))
        assert g.digest(fn(pyorc)) == rec["cases"][name]["out"], name
        names.append(name)
    assert sorted(names) == sorted(rec["cases"])
