"""CPU tests: matcher oracle against straightforward python restatements of the reference loops."""
import numpy as np


def _ham(a, b):
    return int(np.unpackbits(a ^ b).sum())


def _three_max(hist):
    m = [0, 0, 0]; ind = [-1, -1, -1]
    for i, s in enumerate(hist):
        if s > m[0]: m = [s, m[0], m[1]]; ind = [i, ind[0], ind[1]]
        elif s > m[1]: m = [m[0], s, m[1]]; ind = [ind[0], i, ind[1]]
        elif s > m[2]: m[2] = s; ind[2] = i
    if m[1] < np.float32(0.1) * np.float32(m[0]): ind[1] = ind[2] = -1
    elif m[2] < np.float32(0.1) * np.float32(m[0]): ind[2] = -1
    return ind


def _bow_py(variant, d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori):
    n1, n2 = len(d1), len(d2)
    out = np.full(n2 if variant == 0 else n1, -1, np.int64); matched2 = np.zeros(n2, bool)
    bins = {}
    g2 = {int(n): fv2[2][fv2[1][i]:fv2[1][i + 1]] for i, n in enumerate(fv2[0])}
    for i, n in enumerate(fv1[0]):
        if int(n) not in g2: continue
        for idx1 in fv1[2][fv1[1][i]:fv1[1][i + 1]]:
            if not v1[idx1]: continue
            b1, b2, bi = 256, 256, -1
            for idx2 in g2[int(n)]:
                if variant == 0:
                    if out[idx2] >= 0: continue
                else:
                    if matched2[idx2] or not v2[idx2]: continue
                d = _ham(d1[idx1], d2[idx2])
                if d < b1: b2, b1, bi = b1, d, idx2
                elif d < b2: b2 = d
            ok = (b1 <= 50) if variant == 0 else (b1 < 50)
            if ok and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                slot = bi if variant == 0 else idx1
                out[slot] = idx1 if variant == 0 else bi
                matched2[bi] = True
                if ori:
                    rot = np.float32(a1[idx1]) - np.float32(a2[bi])
                    if rot < 0: rot = np.float32(rot + np.float32(360.0))
                    b = int(np.floor(np.float32(rot * np.float32(1.0 / 30)) + np.float32(0.5)))
                    bins[int(slot)] = 0 if b == 30 else b
    if ori:
        hist = [0] * 30
        for b in bins.values(): hist[b] += 1
        keep = _three_max(hist)
        for s, b in bins.items():
            if b not in keep: out[s] = -1
    return out


def _make(rng, synth, n1, n2, nodes):
    base = synth.correlated_descriptors(n1, rng)
    d2, _ = synth.correlated_descriptors(n2, rng, base=base, flip=0.05)
    a1 = rng.uniform(0, 360, n1).astype(np.float32)
    a2 = ((a1[rng.integers(0, n1, n2)] + rng.normal(0, 20, n2)) % 360).astype(np.float32)
    fv1 = synth.feature_vector(n1, nodes, rng); fv2 = synth.feature_vector(n2, nodes, rng)
    v1 = (rng.random(n1) < 0.7).astype(np.uint8); v2 = (rng.random(n2) < 0.8).astype(np.uint8)
    return base, a1, v1, fv1, d2, a2, v2, fv2


def test_bow_oracle_equals_python_loops(pyorc, synth):
    rng = np.random.default_rng(42)
    for trial in range(6):
        n1, n2, nodes = int(rng.integers(50, 400)), int(rng.integers(50, 400)), int(rng.integers(1, 12))
        d1, a1, v1, fv1, d2, a2, v2, fv2 = _make(rng, synth, n1, n2, nodes)
        for variant in (0, 1):
            for ori in (0, 1):
                got, n = pyorc.search_by_bow(variant, d1, a1, v1, pyorc.FeatVec(*fv1), d2, a2, v2, pyorc.FeatVec(*fv2), 0.75, ori)
                exp = _bow_py(variant, d1, a1, v1, fv1, d2, a2, v2, fv2, 0.75, ori)
                assert np.array_equal(got, exp)
                assert n == int((exp >= 0).sum())


def test_triangulation_oracle_properties(pyorc, synth):
    rng = np.random.default_rng(7)
    n1 = n2 = 300
    d1 = synth.correlated_descriptors(n1, rng); d2, src = synth.correlated_descriptors(n2, rng, base=d1, flip=0.04)
    kp1 = np.zeros(n1, pyorc.KP_DTYPE); kp2 = np.zeros(n2, pyorc.KP_DTYPE)
    kp1["x"], kp1["y"] = rng.uniform(0, 1241, n1), rng.uniform(0, 376, n1)
    kp2["x"] = kp1["x"][src] - rng.uniform(0, 40, n2); kp2["y"] = kp1["y"][src] + rng.normal(0, 0.5, n2)
    kp1["angle"] = rng.uniform(0, 360, n1); kp2["angle"] = kp1["angle"][src]
    kp1["octave"] = rng.integers(0, 8, n1); kp2["octave"] = rng.integers(0, 8, n2)
    ur1 = np.where(rng.random(n1) < 0.7, kp1["x"] - 5, -1).astype(np.float32)
    ur2 = np.where(rng.random(n2) < 0.7, kp2["x"] - 5, -1).astype(np.float32)
    mp1 = (rng.random(n1) < 0.3).astype(np.uint8); mp2 = (rng.random(n2) < 0.3).astype(np.uint8)
    node = rng.integers(0, 3, n1)
    def fv(nodes_of):
        ids = np.unique(nodes_of); off = [0]; idx = []
        for i in ids:
            m = np.nonzero(nodes_of == i)[0]; idx.append(m); off.append(off[-1] + len(m))
        return ids.astype(np.uint32), np.asarray(off, np.int32), np.concatenate(idx).astype(np.uint32)
    fv1 = fv(node); fv2 = fv(node[src])
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)     # pure x-translation: epipolar lines are rows
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32); sigma2 = scale * scale
    for only_stereo in (0, 1):
        pairs, n = pyorc.search_for_triangulation(d1, kp1, ur1, mp1, pyorc.FeatVec(*fv1), d2, kp2, ur2, mp2, pyorc.FeatVec(*fv2),
                                                  F12, -1e6, 188.0, scale, sigma2, only_stereo, 1)
        assert n == len(pairs) and n > 10
        assert np.all(np.diff(pairs[:, 0]) > 0)
        for i1, i2 in pairs:
            assert not mp1[i1] and not mp2[i2]
            assert pyorc.descriptor_distance(d1[i1], d2[i2]) <= 50
            assert abs(kp1["y"][i1] - kp2["y"][i2]) ** 2 < 3.84 * sigma2[kp2["octave"][i2]] + 1e-3
            if only_stereo:
                assert ur1[i1] >= 0 and ur2[i2] >= 0
