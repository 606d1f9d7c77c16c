"""GPU parity (bit-exact) of corb_distinctive_descriptors and corb_rebase_map vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_distinctive_descriptors(corb, pyorc):
    rng = np.random.default_rng(11)
    sizes = list(rng.integers(0, 40, 3000)) + [64, 65, 127, 128, 129, 300, 1024, 1, 2, 0]
    offset = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
    rep = np.repeat(np.arange(len(sizes)), sizes)
    bits = np.unpackbits(base[rep], axis=1) ^ (rng.random((len(rep), 256)) < 0.12).astype(np.uint8)
    desc = np.packbits(bits, axis=1)
    g = corb.ComputeDistinctiveDescriptors(desc, offset)
    r = pyorc.distinctive_descriptors(desc, offset)
    assert np.array_equal(g, r)
    # ties: identical descriptors -> the first row
    same = np.repeat(base[:1], 9, 0)
    assert corb.ComputeDistinctiveDescriptors(same, np.array([0, 9], np.int32))[0] == 0


def test_distinctive_descriptors_overflow_is_loud(corb):
    desc = np.zeros((1025, 32), np.uint8)
    with pytest.raises(corb.CorbError):
        corb.ComputeDistinctiveDescriptors(desc, np.array([0, 1025], np.int32))


def test_rebase_map(corb, pyorc):
    rng = np.random.default_rng(12)
    To2n = np.eye(4, dtype=np.float32); A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A); To2n[:3, :3] = Q; To2n[:3, 3] = rng.normal(0, 2, 3)
    poses = rng.normal(0, 1, (1200, 4, 4)).astype(np.float32); pts = rng.normal(0, 20, (48000, 3)).astype(np.float32)
    gP, gX = corb.RebaseMap(To2n, poses, pts)
    rP, rX = pyorc.rebase_map(To2n, poses, pts)
    assert np.array_equal(gP, rP) and np.array_equal(gX, rX)
    gP, gX = corb.RebaseMap(To2n, poses[:0], pts[:5])
    assert len(gP) == 0 and np.array_equal(gX, rX[:5])
