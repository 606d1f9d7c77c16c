"""CPU test of the map push's bookkeeping (corb_map_push_plan, the function corb_map_push_ex runs on the root) behind a mock transport:
counts -> verdict -> dst_first placement -> header invalidation, for N simulated ranks.  No GPU, no communicator: the plan is pure host arithmetic in
libcorb_accel.so; the mock below plays the two all-gathers and the record exchange of corb_comm.cpp on numpy arrays."""
import numpy as np
import pytest


class MockRank:
    def __init__(self, rank, capacity, rec_bytes, rng):
        self.rank = rank; self.capacity = capacity; self.rec_bytes = rec_bytes
        self.records = rng.integers(0, 256, (capacity, rec_bytes), dtype=np.uint8)      # the rank's keyframe store
        self.header_valid = np.ones(capacity, bool)                                     # host mirrors (CorbKfStore::Host::header_valid)


def mock_push(corb, ranks, sends, root, dst_first, local_status=None):
    """the protocol of corb_map_push_ex with the transport replaced by Python lists; returns the per-rank return codes"""
    W = len(ranks)
    hdr = np.zeros(W, corb.PUSH_HEADER_DTYPE)                     # all-gather 1: every rank's header
    for r, rk in enumerate(ranks):
        st = 0 if local_status is None else local_status[r]
        bad_slot = any(s < 0 or s >= rk.capacity for s in sends[r])
        hdr[r] = (st or (-1 if bad_slot else 0), 0 if (st or bad_slot) else len(sends[r]), 0, rk.rec_bytes, 0)
    verdict, who = corb.map_push_plan(hdr, root, ranks[root].capacity, 0, dst_first)   # the root's verdict (all-gather 2 hands it to everybody)
    if verdict != 0:
        return [verdict] * W, who
    staged = [ranks[r].records[list(sends[r])].copy() for r in range(W)]               # every rank packs what it sends
    for r in range(W):                                                                 # the root files rank r's message at dst_first[r]
        n = len(sends[r])
        if n:
            ranks[root].records[dst_first[r]: dst_first[r] + n] = staged[r]
            ranks[root].header_valid[dst_first[r]: dst_first[r] + n] = False
    return [0] * W, -1


def test_placement_and_invalidation(corb):
    rng = np.random.default_rng(1)
    ranks = [MockRank(r, 16, 192, rng) for r in range(4)]
    before = [rk.records.copy() for rk in ranks]
    sends = [[3, 1], [0], [], [5, 6, 7]]
    dst = [8, 10, 11, 11]
    rcs, who = mock_push(corb, ranks, sends, 0, dst)
    assert rcs == [0, 0, 0, 0] and who == -1
    root = ranks[0]
    assert np.array_equal(root.records[8], before[0][3]) and np.array_equal(root.records[9], before[0][1])      # the root's own records, in slot-list order
    assert np.array_equal(root.records[10], before[1][0])
    assert np.array_equal(root.records[11:14], before[3][5:8])
    assert np.array_equal(root.records[:8], before[0][:8]) and np.array_equal(root.records[14:], before[0][14:])  # nothing else is touched
    assert list(np.flatnonzero(~root.header_valid)) == [8, 9, 10, 11, 12, 13]
    for r in (1, 2, 3):
        assert np.array_equal(ranks[r].records, before[r]) and ranks[r].header_valid.all()


def test_every_rank_gets_the_same_error(corb):
    rng = np.random.default_rng(2)
    ranks = [MockRank(r, 8, 128, rng) for r in range(3)]
    before = [rk.records.copy() for rk in ranks]
    # no room on the root for rank 2's records: CORB_ERR_CAPACITY on every rank, nothing moved
    rcs, who = mock_push(corb, ranks, [[0], [1, 2], [0, 1, 2]], 0, [4, 5, 6])
    assert rcs == [-2, -2, -2] and who == 2
    assert all(np.array_equal(rk.records, b) for rk, b in zip(ranks, before))
    # a rank with a bad slot: its status reaches everybody
    rcs, who = mock_push(corb, ranks, [[0], [9], []], 0, [4, 5, 6])
    assert rcs == [-1, -1, -1] and who == 1
    # a rank that rejected its own arguments for any other reason
    rcs, who = mock_push(corb, ranks, [[0], [1], [2]], 0, [4, 5, 6], local_status=[0, 0, -3])
    assert rcs == [-3, -3, -3] and who == 2
    # overlapping destination ranges
    rcs, who = mock_push(corb, ranks, [[0, 1], [1], []], 0, [4, 5, 7])
    assert rcs == [-1, -1, -1] and who == 1
    # destination before the store
    rcs, who = mock_push(corb, ranks, [[0], [], []], 0, [-1, 0, 0])
    assert rcs == [-2, -2, -2] and who == 0


def test_plan_details(corb):
    h = np.zeros(3, corb.PUSH_HEADER_DTYPE)
    h["kf_record_bytes"] = [1024, 2048, 1024]; h["n_kf"] = [1, 1, 1]
    assert corb.map_push_plan(h, 0, 8, 0, [0, 1, 2]) == (-1, 1)               # record sizes differ from the root's (max_features mismatch)
    h["n_kf"] = [1, 0, 1]
    assert corb.map_push_plan(h, 0, 8, 0, [0, 1, 2]) == (0, -1)               # ... but a rank that sends nothing may hold any store
    h["n_mp"] = [0, 0, 5]; h["mp_record_bytes"] = [512, 0, 512]
    assert corb.map_push_plan(h, 0, 8, 4, [0, 1, 2], [0, 0, 0]) == (-2, 2)    # map points do not fit
    assert corb.map_push_plan(h, 0, 8, 5, [0, 1, 2], [0, 0, 0]) == (0, -1)
    assert corb.map_push_plan(h, 0, 8, 5, [0, 1, 2], None)[0] == -1           # map points announced, no destination table
    h["n_kf"] = [1, -1, 1]
    assert corb.map_push_plan(h, 0, 8, 5, [0, 1, 2], [0, 0, 0]) == (-1, 1)
    # an empty push is a valid push; touching ranges are not overlapping ranges
    z = np.zeros(2, corb.PUSH_HEADER_DTYPE)
    assert corb.map_push_plan(z, 1, 1, 0, [0, 0]) == (0, -1)
    z["n_kf"] = [2, 2]; z["kf_record_bytes"] = 64
    assert corb.map_push_plan(z, 1, 4, 0, [2, 0]) == (0, -1)
    assert corb.map_push_plan(z, 1, 4, 0, [1, 0])[0] == -1


def test_messages_of_a_four_rank_push(corb):
    """corb_map_push_messages: what every rank of a 4-rank push posts -- one send per store with records to the root, on the root one receive per rank and store in
    rank order, keyframes before map points, at the destination slots; every send has its receive, in the same order between each pair of ranks."""
    W, root = 4, 1
    hdr = np.zeros(W, corb.PUSH_HEADER_DTYPE)
    hdr["n_kf"] = [2, 1, 0, 3]; hdr["n_mp"] = [40, 0, 7, 0]; hdr["kf_record_bytes"] = 1024; hdr["mp_record_bytes"] = 320
    kd = [0, 10, 20, 30]; md = [0, 100, 200, 300]
    posted = {}
    for r in range(W):
        sends, recvs = corb.map_push_messages(hdr, r, root, kd if r == root else None, md if r == root else None)
        posted[r] = (sends, recvs)
        assert [int(m["kind"]) for m in sends] == ([0] if hdr["n_kf"][r] else []) + ([1] if hdr["n_mp"][r] else [])
        assert all(int(m["peer"]) == root for m in sends)
        assert [int(m["bytes"]) for m in sends] == [int(hdr["n_kf"][r]) * 1024] * bool(hdr["n_kf"][r]) + [int(hdr["n_mp"][r]) * 320] * bool(hdr["n_mp"][r])
        if r != root:
            assert len(recvs) == 0
    recvs = posted[root][1]
    assert [(int(m["peer"]), int(m["kind"]), int(m["first_record"]), int(m["n_records"])) for m in recvs] == [(0, 0, 0, 2), (0, 1, 0, 40), (1, 0, 10, 1), (2, 1, 200, 7), (3, 0, 30, 3)]
    for r in range(W):                                             # pairwise matching in posting order
        from_r = [(int(m["kind"]), int(m["bytes"])) for m in recvs if int(m["peer"]) == r]
        assert from_r == [(int(m["kind"]), int(m["bytes"])) for m in posted[r][0]]
    with pytest.raises(RuntimeError):
        corb.map_push_messages(hdr, root, root, None, md)           # a root without its destination table


def test_rccl_branch_posts_the_messages_in_one_group(corb):
    """the RCCL branch of the record exchange on a recording fake (corb_comm_test_rccl_exchange): ncclGroupStart, the sends, the receives, ncclGroupEnd -- byte counts
    as announced, datatype ncclInt8 -- and a failing ncclSend still closes the group and returns an error (round 3: an early return left the group open)."""
    W, root = 4, 0
    hdr = np.zeros(W, corb.PUSH_HEADER_DTYPE)
    hdr["n_kf"] = [1, 2, 3, 4]; hdr["n_mp"] = [16, 0, 16, 16]; hdr["kf_record_bytes"] = 2048; hdr["mp_record_bytes"] = 320
    sends, recvs = corb.map_push_messages(hdr, root, root, [0, 8, 16, 24], [0, 64, 128, 192])
    rc, log = corb.rccl_exchange_with_fake(sends, recvs)
    assert rc == 0
    assert log[0] == ("group_start",) and log[-1] == ("group_end",) and sum(1 for e in log if e[0].startswith("group")) == 2
    body = log[1:-1]
    assert [e[0] for e in body] == ["send"] * len(sends) + ["recv"] * len(recvs)
    assert [(e[1], e[2]) for e in body if e[0] == "send"] == [(root, 2048), (root, 16 * 320)]
    assert [(e[1], e[2]) for e in body if e[0] == "recv"] == [(0, 2048), (0, 5120), (1, 4096), (2, 6144), (2, 5120), (3, 8192), (3, 5120)]
    assert all(e[3] == 0 for e in body)                           # ncclInt8: counts are bytes
    rc, log = corb.rccl_exchange_with_fake(sends, recvs, fail_at=1)
    assert rc != 0 and log[0] == ("group_start",) and log[-1] == ("group_end",) and len(log) == 4      # start, send ok, send fails, end: nothing posted after the failure
    sends2, recvs2 = corb.map_push_messages(hdr, 2, root)
    rc, log = corb.rccl_exchange_with_fake(sends2, recvs2)
    assert rc == 0 and [e[0] for e in log] == ["group_start", "send", "send", "group_end"]
