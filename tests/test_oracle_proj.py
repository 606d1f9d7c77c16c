"""CPU tests: projection-matcher oracle against straightforward python loops of the reference code."""
import numpy as np


def _ham(a, b):
    return int(np.unpackbits(a ^ b).sum())


def _grid(cur):
    winv = np.float32(64) / (np.float32(cur["max_x"]) - np.float32(cur["min_x"])); hinv = np.float32(48) / (np.float32(cur["max_y"]) - np.float32(cur["min_y"]))
    cells = [[[] for _ in range(48)] for _ in range(64)]
    for i, k in enumerate(cur["keys_un"]):
        px = int(np.floor((np.float32(k["x"]) - np.float32(cur["min_x"])) * winv + np.float32(0.5)))      # round() for non-negative values
        py = int(np.floor((np.float32(k["y"]) - np.float32(cur["min_y"])) * hinv + np.float32(0.5)))
        if 0 <= px < 64 and 0 <= py < 48: cells[px][py].append(i)
    return cells, winv, hinv


def _area(cur, cells, winv, hinv, x, y, r, lo, hi):
    x, y, r = np.float32(x), np.float32(y), np.float32(r)
    x0 = max(0, int(np.floor((x - np.float32(cur["min_x"]) - r) * winv))); x1 = min(63, int(np.ceil((x - np.float32(cur["min_x"]) + r) * winv)))
    y0 = max(0, int(np.floor((y - np.float32(cur["min_y"]) - r) * hinv))); y1 = min(47, int(np.ceil((y - np.float32(cur["min_y"]) + r) * hinv)))
    if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0: return []
    chk = lo > 0 or hi >= 0
    out = []
    for ix in range(x0, x1 + 1):
        for iy in range(y0, y1 + 1):
            for f in cells[ix][iy]:
                k = cur["keys_un"][f]
                if chk and (k["octave"] < lo or (hi >= 0 and k["octave"] > hi)): continue
                if abs(np.float32(k["x"]) - x) < r and abs(np.float32(k["y"]) - y) < r: out.append(f)
    return out


def test_projection_map_oracle_equals_python(pyorc, synth):
    s = synth.tracking_scene(seed=4001, n=700)
    cur, mps, desc = s["cur"], s["mps"], s["last_desc"]
    got, n = pyorc.search_by_projection_map(cur, mps, desc, 3.0, 0.8)
    cells, winv, hinv = _grid(cur)
    claimed = cur["claimed"].astype(bool).copy(); exp = np.full(len(cur["keys_un"]), -1); nm = 0
    for i, p in enumerate(mps):
        if not p["valid"]: continue
        r = np.float32(2.5 if float(p["view_cos"]) > 0.998 else 4.0) * np.float32(3.0)
        win = r * cur["scale"][p["level"]]
        b1, b2, l1, l2, bi = 256, 256, -1, -1, -1
        for f in _area(cur, cells, winv, hinv, p["proj_x"], p["proj_y"], win, p["level"] - 1, p["level"]):
            if claimed[f]: continue
            if cur["u_right"][f] > 0 and abs(np.float32(p["proj_xr"]) - cur["u_right"][f]) > win: continue
            d = _ham(desc[i], cur["desc"][f])
            if d < b1: b2, l2, b1, l1, bi = b1, l1, d, cur["keys_un"][f]["octave"], f
            elif d < b2: b2, l2 = d, cur["keys_un"][f]["octave"]
        if b1 <= 100:
            if l1 == l2 and np.float32(b1) > np.float32(0.8) * np.float32(b2): continue
            exp[bi] = i; claimed[bi] = bool(p["claims"]); nm += 1
    assert n == nm and np.array_equal(got, exp) and nm > 50


def test_projection_frame_oracle_properties(pyorc, synth):
    for seed, motion in ((4002, (0, 0, 0.8)), (4003, (0, 0, -0.9)), (4004, (0.3, 0, 0.1))):
        s = synth.tracking_scene(seed=seed, n=900, motion=motion)
        for mono in (0, 1):
            got, n = pyorc.search_by_projection_frame(s["cur"], s["Tcw"], s["Tlw"], s["fx"], s["fy"], s["cx"], s["cy"], s["bf"], s["mb"],
                                                      s["last"], s["last_desc"], 15.0 if mono else 7.0, mono, 1)
            assert n > 100
            idx = np.nonzero(got >= 0)[0]
            assert len(idx) <= n                                  # events can exceed distinct features only through re-assignment
            for f in idx[:200]:
                q = got[f]
                assert s["last"][q]["valid"] and not s["cur"]["claimed"][f]
                assert _ham(s["last_desc"][q], s["cur"]["desc"][f]) <= 100
