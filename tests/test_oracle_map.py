"""Oracle checks for ComputeDistinctiveDescriptors and the server map re-basing against direct numpy restatements."""
import numpy as np


def _popcount(a):
    return np.unpackbits(a, axis=-1).sum(-1)


def test_distinctive_descriptors_vs_numpy(pyorc):
    rng = np.random.default_rng(7)
    sizes = [1, 2, 3, 4, 7, 10, 33, 64, 65, 130, 0, 5]
    offset = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
    desc = np.zeros((offset[-1], 32), np.uint8)
    for p, n in enumerate(sizes):
        bits = np.repeat(np.unpackbits(base[p])[None], n, 0) ^ (rng.random((n, 256)) < 0.1).astype(np.uint8)
        desc[offset[p]:offset[p + 1]] = np.packbits(bits, axis=1)
    got = pyorc.distinctive_descriptors(desc, offset)
    for p, n in enumerate(sizes):
        if n == 0:
            assert got[p] == -1; continue
        D = desc[offset[p]:offset[p + 1]]
        dist = _popcount(D[:, None, :] ^ D[None, :, :])
        med = np.sort(dist, axis=1)[:, int(0.5 * (n - 1))]
        assert got[p] == int(np.argmin(med))                        # argmin = first minimum, like the strict '<'


def test_rebase_map_vs_numpy(pyorc):
    rng = np.random.default_rng(8)
    def pose():
        A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A); T = np.eye(4); T[:3, :3] = Q * np.sign(np.linalg.det(Q)); T[:3, 3] = rng.normal(0, 3, 3); return T
    To2n = pose().astype(np.float32)
    poses = np.stack([pose() for _ in range(50)]).astype(np.float32); pts = rng.normal(0, 10, (400, 3)).astype(np.float32)
    P, X = pyorc.rebase_map(To2n, poses, pts)
    assert np.array_equal(P, (poses.astype(np.float64) @ To2n.astype(np.float64)).astype(np.float32))
    d = pts - To2n[:3, 3]
    assert np.array_equal(X, (d.astype(np.float64) @ To2n[:3, :3].astype(np.float64)).astype(np.float32))
    # a point expressed in the old map and re-based lands where the composed pose sees it: Tcw p == (Tcw To2n) p'
    p = np.append(pts[0].astype(np.float64), 1.0); pn = np.append(X[0].astype(np.float64), 1.0)
    assert np.allclose(poses[0].astype(np.float64) @ p, P[0].astype(np.float64) @ pn, atol=1e-4)


def test_mappoint_replace_oracle_equals_a_dict_model(pyorc):
    """orc_mappoint_replace (MapPoint.cc:277-316 on flat lists) against the obvious model: std::map semantics with Python dicts, 200 random list pairs"""
    rng = np.random.default_rng(5)
    for case in range(200):
        cap = int(rng.integers(1, 10))
        na, nb = int(rng.integers(0, cap + 1)), int(rng.integers(0, cap + 1))
        ka = sorted(rng.choice(14, na, replace=False).tolist()); kb = sorted(rng.choice(14, nb, replace=False).tolist())
        oa = [(k, int(rng.integers(0, 99))) for k in ka]; ob = [(k, int(rng.integers(0, 99))) for k in kb]
        st, into, act, cnt = pyorc.mappoint_replace(7, 9, oa, ob, cap, (3, 4), (10, 20))
        d = dict(ob); want_act = []
        for k, idx in oa:                                               # :300-313
            if k not in d: d[k] = idx; want_act.append(1)
            else: want_act.append(2)
        if len(d) > cap:
            assert st == -1 and into == ob
            continue
        assert st == 0 and into == sorted(d.items()) and list(act) == want_act and cnt == (13, 24)
    assert pyorc.mappoint_replace(7, 7, [(1, 1)], [(2, 2)], 4)[0] == 1
