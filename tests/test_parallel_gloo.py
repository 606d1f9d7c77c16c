"""CPU test of the N>1 path: world_size=2, gloo, 127.0.0.1 -- timing reduction and the keyframe gather."""
import os
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import corbload
    corbload.load_pkg()
    from corb_slam_amd import parallel
    t, u = parallel.reduce_step_time(dist, 1.0 + rank, 100.0 * (rank + 1))
    rng = np.random.default_rng(rank)
    n = 5 + 3 * rank
    kp = rng.integers(0, 256, (n, 28), dtype=np.uint8); desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ur = rng.random(n).astype(np.float32)
    got = parallel.gather_keyframes(dist, kp, desc, ur, dst=0)
    ok = True
    if rank == 0:
        ok = len(got) == world
        for r in range(world):
            g = np.random.default_rng(r); m = 5 + 3 * r
            ekp = g.integers(0, 256, (m, 28), dtype=np.uint8); ed = g.integers(0, 256, (m, 32), dtype=np.uint8); eu = g.random(m).astype(np.float32)
            ok = ok and np.array_equal(got[r][0], ekp) and np.array_equal(got[r][1], ed) and np.array_equal(got[r][2], eu)
    else:
        ok = got is None
    q.put((rank, t, u, ok, parallel.client_frame_offset(rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduction_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, t, u, ok, off in res:
        assert t == 2.0 and u == 300.0 and ok            # MAX over ranks of time, SUM of units
        assert off == 64 * rank
