"""GPU parity tests (pytest -m gpu): HIP extraction + stereo matching vs the CPU oracle, through the
C-ABI.  Bar: bit-exact for every integer AND float output (the float paths are specified as non-fused
IEEE sequences on both sides, so equality is exact, tolerance 0)."""
import hashlib
import json
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _same_kps(a, b):
    assert len(a) == len(b)
    assert a.tobytes() == b.tobytes(), "keypoints differ"


@pytest.fixture(scope="module")
def ex(corb):
    e = corb.ORBextractor()
    yield e
    e.close()


def test_tables(corb, pyorc, ex):
    g, r = ex.tables(), pyorc.Extractor().tables()
    for k in g:
        assert np.array_equal(g[k], r[k]), k


@pytest.mark.parametrize("idx", [0, 1, 5])
def test_extract_stages_bit_exact(corb, pyorc, synth, ex, idx):
    L, _ = synth.stereo_pair(idx)
    kps, desc = ex(L)
    ref = pyorc.Extractor(); rk, rd = ref.extract(L)
    for l in range(8):
        assert np.array_equal(ex.pyramid_level(0, l), ref.level(l)), "pyramid level %d" % l
        if ref.blurred(l) is not None:
            assert np.array_equal(ex.pyramid_level(0, l, True), ref.blurred(l)), "blur level %d" % l
        g, r = ex.candidates(0, l), ref.candidates(l)
        assert len(g) == len(r) and all(np.array_equal(g[f], r[f]) for f in ("x", "y", "response")), "FAST level %d" % l
    _same_kps(kps, rk)
    assert np.array_equal(desc, rd)


def test_extract_edge_cases(corb, pyorc, synth, ex):
    k, d = ex(np.zeros((0, 0), np.uint8))                  # empty image -> n = 0, success
    assert len(k) == 0
    k, d = ex(synth.flat_image(1241, 376))                 # no corners anywhere
    assert len(k) == 0 and d.shape == (0, 32)
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (376, 1241), dtype=np.uint8)      # maximum candidate density
    k, d = ex(noise)
    rk, rd = pyorc.Extractor().extract(noise)
    _same_kps(k, rk); assert np.array_equal(d, rd)
    img = synth.flat_image(1241, 376); img[100:140, 300:360] = 255; img[200:203, 800:803] = 0   # few corners, fallback cells
    k, d = ex(img); rk, rd = pyorc.Extractor().extract(img)
    _same_kps(k, rk); assert np.array_equal(d, rd)
    with pytest.raises(corb.CorbError):
        ex(np.zeros((100, 100), np.uint8))                # size mismatch with the handle -> loud error


@pytest.mark.parametrize("cfg", [dict(nfeatures=500, scaleFactor=1.5, nlevels=4, iniThFAST=30, minThFAST=10, width=320, height=240),
                                 dict(nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, width=752, height=480),
                                 dict(nfeatures=4000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, width=1920, height=1080),
                                 # odd sizes: partial 4-px groups and reflected columns at both image borders on every level, 1-3 row strips
                                 dict(nfeatures=300, scaleFactor=1.2, nlevels=3, iniThFAST=20, minThFAST=7, width=129, height=97),
                                 dict(nfeatures=800, scaleFactor=1.3, nlevels=5, iniThFAST=15, minThFAST=5, width=403, height=263),
                                 dict(nfeatures=1500, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, width=1283, height=381)])
def test_other_configurations(corb, pyorc, synth, cfg):
    L, _ = synth.stereo_pair(3, cfg["width"], cfg["height"])
    e = corb.ORBextractor(**cfg)
    k, d = e(L)
    ref = pyorc.Extractor(cfg["nfeatures"], cfg["scaleFactor"], cfg["nlevels"], cfg["iniThFAST"], cfg["minThFAST"])
    rk, rd = ref.extract(L)
    for l in range(cfg["nlevels"]):
        assert np.array_equal(e.pyramid_level(0, l), ref.level(l)), "pyramid level %d" % l
        if ref.blurred(l) is not None:
            assert np.array_equal(e.pyramid_level(0, l, True), ref.blurred(l)), "blur level %d" % l
    _same_kps(k, rk); assert np.array_equal(d, rd)
    e.close()


def test_stereo_batch_matches_oracle_and_golden(corb, pyorc, synth):
    g = json.load(open(os.path.join(GOLD, "orb_stereo_kitti.json")))
    frames = [r["frame"] for r in g["frames"]]
    sf = corb.StereoFrontend(max_frames=len(frames))
    for s, f in enumerate(frames):
        l, r = synth.stereo_pair(f); sf.upload(s, l, r)
    sf.run(len(frames)); sf.sync()
    for s, rec in enumerate(g["frames"]):
        out = sf.fetch(s)
        assert (len(out["kl"]), len(out["kr"]), out["n_matched"]) == (rec["n_left"], rec["n_right"], rec["n_matched"])
        assert sha(out["kl"]) == rec["sha"]["kp_left"] and sha(out["dl"]) == rec["sha"]["desc_left"]
        assert sha(out["kr"]) == rec["sha"]["kp_right"] and sha(out["dr"]) == rec["sha"]["desc_right"]
        assert sha(out["u_right"]) == rec["sha"]["u_right"] and sha(out["depth"]) == rec["sha"]["depth"]
    # live oracle on one more frame
    l, r = synth.stereo_pair(11); sf.upload(0, l, r); sf.run(1); sf.sync(); out = sf.fetch(0)
    el, er = pyorc.Extractor(), pyorc.Extractor()
    kl, dl = el.extract(l); kr, dr = er.extract(r); tb = el.tables()
    ur, dp, nm = pyorc.stereo_match(el, er, kl, dl, kr, dr, 386.1448, 718.856, tb["scale"], tb["inv_scale"])
    _same_kps(out["kl"], kl); _same_kps(out["kr"], kr)
    assert np.array_equal(out["u_right"].view(np.uint32), ur.view(np.uint32))
    assert np.array_equal(out["depth"].view(np.uint32), dp.view(np.uint32)) and out["n_matched"] == nm
    sf.close()


def test_stereo_frames_call_matches_golden_and_oracle(corb, pyorc, synth):
    """corb_stereo_frames -- Frame::Frame(stereo) as a client calls it, one frame (or a few) per call, host buffers in and out, one transfer each way around
    the captured kernel chain: the golden SHA fixtures frame by frame at n = 1, the same frames at n = 2 and n = 3 (another captured graph each, blocks of a
    multi-frame result), pageable and page-locked buffers, the live oracle on an edge case (no keypoints in the right eye) after a dense frame has used
    the same result block (stale entries beyond the counts must not show)."""
    g = json.load(open(os.path.join(GOLD, "orb_stereo_kitti.json")))
    recs = g["frames"][:6]
    sf = corb.StereoFrontend(max_frames=4)
    lay = sf.frame_layout()
    assert lay.capacity == corb.load().corb_orb_capacity(sf.orb.h) and lay.frame_bytes % 64 == 0 and lay.off_kp_left == 64

    def check(o, rec):
        assert (len(o["kl"]), len(o["kr"]), o["n_matched"], o.get("status", 0)) == (rec["n_left"], rec["n_right"], rec["n_matched"], 0)
        assert sha(o["kl"]) == rec["sha"]["kp_left"] and sha(o["dl"]) == rec["sha"]["desc_left"]
        assert sha(o["kr"]) == rec["sha"]["kp_right"] and sha(o["dr"]) == rec["sha"]["desc_right"]
        assert sha(o["u_right"]) == rec["sha"]["u_right"] and sha(o["depth"]) == rec["sha"]["depth"]

    pairs = [np.stack(synth.stereo_pair(r["frame"])) for r in recs]
    for rec, p in zip(recs, pairs):                                   # n = 1, pageable buffers
        check(sf.unpack_frame(sf.frames(p[None])), rec)
    pin_in = corb.pinned_empty((3,) + pairs[0].shape, np.uint8); pin_out = corb.pinned_empty((3 * lay.frame_bytes,), np.uint8)
    tm = corb.StereoFrameTiming()
    for n in (2, 3, 1, 3):                                            # page-locked buffers, several frames per call, the graphs of n = 1, 2, 3 in turn
        for i0 in range(0, len(recs) - n + 1, n):
            for k in range(n): pin_in[k] = pairs[i0 + k]
            res = sf.frames(pin_in[:n], pin_out, tm)
            for k in range(n): check(sf.unpack_frame(res, k), recs[i0 + k])
            assert tm.ms_upload > 0 and tm.ms_kernels > 0 and tm.ms_download > 0
    # an eye without keypoints, in the block a dense frame has just filled
    l, _ = synth.stereo_pair(3); flat = synth.flat_image(1241, 376)
    o = sf.unpack_frame(sf.frames(np.stack([l, flat])[None]))
    el, er = pyorc.Extractor(), pyorc.Extractor()
    kl, dl = el.extract(l); kr, dr = er.extract(flat); tb = el.tables()
    ur, dp, nm = pyorc.stereo_match(el, er, kl, dl, kr, dr, 386.1448, 718.856, tb["scale"], tb["inv_scale"])
    _same_kps(o["kl"], kl); assert len(o["kr"]) == len(kr) == 0 and o["n_matched"] == nm == 0
    assert np.array_equal(o["dl"], dl) and np.array_equal(o["u_right"].view(np.uint32), ur.view(np.uint32)) and np.array_equal(o["depth"].view(np.uint32), dp.view(np.uint32))
    # the batch entry points still serve the handle afterwards (the frames call is unsplit and joins like every other call)
    sf.upload(0, *synth.stereo_pair(recs[0]["frame"])); sf.run(1); sf.sync(); check(sf.fetch(0), recs[0])
    sf.close()
    # a batch upload that outgrows the staging buffer the captured chains read from: the next per-frame call captures them again
    sf = corb.StereoFrontend(max_frames=12)
    check(sf.unpack_frame(sf.frames(pairs[0][None])), recs[0])
    sf.upload_batch(0, np.ascontiguousarray(np.stack([pairs[i % 3] for i in range(12)]))); sf.run(12); sf.sync()
    check(sf.fetch(11), recs[11 % 3])
    check(sf.unpack_frame(sf.frames(pairs[1][None])), recs[1])
    sf.close()


def test_stereo_edge_cases_match_oracle(corb, pyorc, synth):
    """frames without keypoints, with a handful of matches, with unrelated eyes (few / no accepted matches) and a maximum-density pair: the row table,
    the matcher and the median filter against the oracle, in one batch"""
    rng = np.random.default_rng(5)
    flat = synth.flat_image(1241, 376)
    few = flat.copy(); few[100:140, 300:360] = 255; few[200:230, 800:830] = 30
    few_r = np.roll(few, -9, axis=1)
    l3, _ = synth.stereo_pair(3); _, r9 = synth.stereo_pair(9)
    noise = rng.integers(0, 256, (376, 1241), dtype=np.uint8)
    cases = [(flat, flat), (few, few_r), (l3, r9), (noise, np.roll(noise, -5, axis=1)), (l3, flat), (flat, r9)]
    sf = corb.StereoFrontend(max_frames=len(cases))
    for s, (l, r) in enumerate(cases): sf.upload(s, l, r)
    sf.run(len(cases)); sf.sync()
    for s, (l, r) in enumerate(cases):
        out = sf.fetch(s)
        el, er = pyorc.Extractor(), pyorc.Extractor()
        kl, dl = el.extract(l); kr, dr = er.extract(r); tb = el.tables()
        ur, dp, nm = pyorc.stereo_match(el, er, kl, dl, kr, dr, 386.1448, 718.856, tb["scale"], tb["inv_scale"])
        _same_kps(out["kl"], kl); _same_kps(out["kr"], kr)
        assert np.array_equal(out["u_right"].view(np.uint32), ur.view(np.uint32)), "case %d" % s
        assert np.array_equal(out["depth"].view(np.uint32), dp.view(np.uint32)) and out["n_matched"] == nm, "case %d" % s
    sf.close()


def test_stereo_1080p_golden(corb, synth):
    g = json.load(open(os.path.join(GOLD, "orb_stereo_1080p.json")))
    rec = g["frames"][0]
    sf = corb.StereoFrontend(nfeatures=rec["nfeatures"], width=1920, height=1080, max_frames=1, fx=rec["fx"], bf=rec["bf"])
    l, r = synth.stereo_pair(rec["frame"], 1920, 1080); sf.upload(0, l, r); sf.run(1); sf.sync(); out = sf.fetch(0)
    assert (len(out["kl"]), len(out["kr"]), out["n_matched"]) == (rec["n_left"], rec["n_right"], rec["n_matched"])
    assert sha(out["kl"]) == rec["sha"]["kp_left"] and sha(out["dl"]) == rec["sha"]["desc_left"]
    assert sha(out["u_right"]) == rec["sha"]["u_right"] and sha(out["depth"]) == rec["sha"]["depth"]
    sf.close()


def test_full_batch_properties(corb, synth):
    """Bench-size batch (32 frames): size-independent properties -- batch slots are independent
    (same image in two slots -> identical bytes), re-running is idempotent, frame order does not matter."""
    B = 32
    sf = corb.StereoFrontend(max_frames=B)
    pairs = [synth.stereo_pair(i % 8) for i in range(B)]
    for s, (l, r) in enumerate(pairs): sf.upload(s, l, r)
    sf.run(B); sf.sync()
    outs = [sf.fetch(s) for s in range(B)]
    for s in range(8, B):
        a, b = outs[s], outs[s % 8]
        for k in ("kl", "dl", "kr", "dr", "u_right", "depth"):
            assert a[k].tobytes() == b[k].tobytes(), (s, k)
    sf.run(B); sf.sync()
    again = sf.fetch(5)
    assert all(again[k].tobytes() == outs[5][k].tobytes() for k in ("kl", "dl", "u_right", "depth"))
    # left keypoints with a stereo match have 0 <= disparity < fx and depth = bf/disparity
    o = outs[3]; m = o["u_right"] >= 0
    disp = o["kl"]["x"][m] - o["u_right"][m]
    assert m.sum() == o["n_matched"] and np.all(disp > 0) and np.all(disp < 718.9)
    sf.close()


def test_many_frames_match_oracle(corb, pyorc, synth):
    """a wider sweep than the fixtures: 20 more frames of different texture contrast and rectangle density (dense and sparse FAST lists, cells that
    need the second threshold, rows with many and with few stereo candidates), processed as ONE batch (the part-batch path) and compared with the
    oracle frame by frame -- keypoints, descriptors, mvuRight / mvDepth as bit patterns"""
    rng = np.random.default_rng(77)
    frames = []
    for i in range(20):
        l, r = synth.stereo_pair(500 + i, n_rect=int(rng.integers(20, 900)), contrast=float(rng.uniform(0.05, 2.5)))
        if i % 5 == 4:                                        # a dark / low-contrast eye: few keypoints on one side
            r = (r.astype(np.float32) * 0.25 + 90).astype(np.uint8)
        frames.append((l, r))
    sf = corb.StereoFrontend(max_frames=len(frames))
    for s, (l, r) in enumerate(frames): sf.upload(s, l, r)
    sf.run(len(frames)); sf.sync()
    for s, (l, r) in enumerate(frames):
        out = sf.fetch(s)
        el, er = pyorc.Extractor(), pyorc.Extractor()
        kl, dl = el.extract(l); kr, dr = er.extract(r); tb = el.tables()
        ur, dp, nm = pyorc.stereo_match(el, er, kl, dl, kr, dr, 386.1448, 718.856, tb["scale"], tb["inv_scale"])
        _same_kps(out["kl"], kl); _same_kps(out["kr"], kr)
        assert np.array_equal(out["dl"], dl) and np.array_equal(out["dr"], dr), "descriptors, frame %d" % s
        assert np.array_equal(out["u_right"].view(np.uint32), ur.view(np.uint32)), "mvuRight, frame %d" % s
        assert np.array_equal(out["depth"].view(np.uint32), dp.view(np.uint32)) and out["n_matched"] == nm, "mvDepth, frame %d" % s
    sf.close()
