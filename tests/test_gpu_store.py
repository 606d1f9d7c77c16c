"""GPU tests: device-resident keyframe store, matchers on store slots, RCCL map push (include/corb_accel.h, corb_kf_store_* / corb_map_push)."""
import numpy as np
import pytest
from test_oracle_match import _make

pytestmark = pytest.mark.gpu


def test_slot_filled_on_the_device_equals_the_fetched_results(corb, synth):
    """corb_kf_store_put_from_stereo: the left keypoints / descriptors / mvuRight / mvDepth of a frame go into a slot device-to-device"""
    sf = corb.StereoFrontend(max_frames=2)
    for i in range(2):
        l, r = synth.stereo_pair(70 + i); sf.upload(i, l, r)
    sf.run(2)
    st = corb.KeyFrameStore(4, corb.load().corb_orb_capacity(sf.orb.h))
    st.put_from_stereo(2, sf, 1, keyframe_id=4242); st.put_from_stereo(0, sf, 0, keyframe_id=7)
    sf.sync()
    for slot, frame, kid in ((2, 1, 4242), (0, 0, 7)):
        ref = sf.fetch(frame); got = st.get(slot)
        assert got["id"] == kid and len(got["kp"]) == len(ref["kl"]) > 1000
        assert got["kp"].tobytes() == ref["kl"].tobytes() and np.array_equal(got["desc"], ref["dl"])
        assert np.array_equal(got["u_right"].view(np.uint32), ref["u_right"].view(np.uint32)) and np.array_equal(got["depth"].view(np.uint32), ref["depth"].view(np.uint32))
        assert not got["flags"].any() and len(got["fv"][0]) == 0
    assert st.record_bytes() % 64 == 0
    st.close(); sf.close()


def test_matchers_on_slots_equal_the_host_pointer_calls(corb, pyorc, synth):
    rng = np.random.default_rng(9)
    d1, a1, v1, fv1, d2, a2, v2, fv2 = _make(rng, synth, 900, 800, 40)
    n1, n2 = 900, 800
    kp1 = np.zeros(n1, corb.KP_DTYPE); kp2 = np.zeros(n2, corb.KP_DTYPE)
    kp1["angle"] = a1; kp2["angle"] = a2
    kp1["x"], kp1["y"] = rng.uniform(0, 1241, n1), rng.uniform(0, 376, n1); kp2["x"], kp2["y"] = rng.uniform(0, 1241, n2), rng.uniform(0, 376, n2)
    kp1["octave"] = rng.integers(0, 8, n1); kp2["octave"] = rng.integers(0, 8, n2)
    ur1 = np.where(rng.random(n1) < 0.6, kp1["x"] - 5, -1).astype(np.float32); ur2 = np.where(rng.random(n2) < 0.6, kp2["x"] - 5, -1).astype(np.float32)
    A = corb.KeyFrameStore(3, 1024); B = corb.KeyFrameStore(2, 1024)
    A.put(1, kp1, d1, ur1, None, keyframe_id=11); A.set_bow(1, fv1); A.set_flags(1, v1)
    B.put(0, kp2, d2, ur2, None, keyframe_id=22); B.set_bow(0, fv2); B.set_flags(0, v2)
    g = A.get(1)
    assert np.array_equal(g["flags"], v1) and all(np.array_equal(x, y) for x, y in zip(g["fv"], (fv1[0], fv1[1], fv1[2])))
    for variant in (0, 1):
        for ratio, ori in ((0.9, True), (0.75, False)):
            m, n = A.SearchByBoW(1, B, 0, ratio, ori, variant)
            r, rn = pyorc.search_by_bow(variant, d1, a1, v1, pyorc.FeatVec(*fv1), d2, a2, v2 if variant == 1 else np.ones_like(v2), pyorc.FeatVec(*fv2), ratio, ori)
            assert n == rn and np.array_equal(m, r)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-4, (3, 3)).astype(np.float32)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32); sigma2 = scale * scale
    for only_stereo in (False, True):
        gp, gn = A.SearchForTriangulation(1, B, 0, F12, 600.0, 180.0, scale, sigma2, only_stereo)
        rp, rn = pyorc.search_for_triangulation(d1, kp1, ur1, v1, pyorc.FeatVec(*fv1), d2, kp2, ur2, v2, pyorc.FeatVec(*fv2), F12, 600.0, 180.0, scale, sigma2, only_stereo, True)
        assert gn == rn and np.array_equal(gp.reshape(-1, 2), np.asarray(rp).reshape(-1, 2))
    # an empty slot / a slot without BoW groups matches nothing
    m, n = A.SearchByBoW(0, B, 0)
    assert n == 0
    with pytest.raises(corb.CorbError):
        A.put(0, np.zeros(2000, corb.KP_DTYPE), np.zeros((2000, 32), np.uint8))           # more features than the store holds per keyframe
    A.close(); B.close()


def test_map_push_over_rccl_single_rank(corb, synth):
    """corb_map_push with a one-rank communicator (the GPU box has one GPU): the records travel through ncclSend / ncclRecv on the device buffers
    (rank 0 sends to itself) and arrive bit-identical, BoW groups and flags included; the N-rank path is the same code with more peers."""
    rng = np.random.default_rng(10)
    st = corb.KeyFrameStore(8, 512)
    ref = []
    for s in range(3):
        n = 300 + 50 * s
        kp = np.zeros(n, corb.KP_DTYPE); kp["x"] = rng.uniform(0, 1000, n); kp["angle"] = rng.uniform(0, 360, n); kp["octave"] = rng.integers(0, 8, n)
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8); ur = rng.uniform(-1, 900, n).astype(np.float32); dp = rng.uniform(1, 50, n).astype(np.float32)
        fv = synth.feature_vector(n, 12, rng); fl = (rng.random(n) < 0.5).astype(np.uint8)
        st.put(s, kp, desc, ur, dp, keyframe_id=1000 + s); st.set_bow(s, fv); st.set_flags(s, fl)
        ref.append((kp, desc, ur, dp, fv, fl))
    comm = corb.Comm(corb.Comm.unique_id(), 0, 1)
    cnt = comm.map_push(st, [0, 2], root=0, dst_first=[5])
    assert list(cnt) == [2]
    for dst, src in ((5, 0), (6, 2)):
        g = st.get(dst); kp, desc, ur, dp, fv, fl = ref[src]
        assert g["id"] == 1000 + src and g["kp"].tobytes() == kp.tobytes() and np.array_equal(g["desc"], desc)
        assert np.array_equal(g["u_right"], ur) and np.array_equal(g["depth"], dp) and np.array_equal(g["flags"], fl)
        assert all(np.array_equal(x, np.asarray(y)) for x, y in zip(g["fv"], fv))
    assert list(comm.map_push(st, [], root=0, dst_first=[7])) == [0]                      # nothing new to push is a valid push
    with pytest.raises(corb.CorbError):
        comm.map_push(st, [0, 1, 2], root=0, dst_first=[6])                                # no room at the destination
    # the pushed copy is matchable like the original
    m1, n1 = st.SearchByBoW(0, st, 1, 0.9, True, 1); m2, n2 = st.SearchByBoW(5, st, 1, 0.9, True, 1)
    assert n1 == n2 and np.array_equal(m1, m2)
    comm.close(); st.close()
