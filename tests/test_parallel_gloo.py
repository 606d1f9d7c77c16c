"""CPU test of bench.py's N>1 path: world_size=2, gloo, 127.0.0.1 -- the timing reductions (MAX time, SUM frames, per-rank figures).
The map push is not a torch.distributed path: see tests/test_push_plan.py (CPU) and tests/test_gpu_mapstore.py."""
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
    per = parallel.gather_scalars(dist, 10.0 * (rank + 1))
    ok = per == [10.0, 20.0]
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


def test_bench_gpus_flag_starts_that_many_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher must start 2 ranks itself (round 2 parsed the flag and ran one).  No GPU here: the ranks stop with the
    one-client-per-GPU message and the exit code is non-zero (no silent 1-GPU run).  That two ranks were started is read from the marker file each rank
    leaves as its first action -- not from counting stderr messages: torchrun SIGTERMs the surviving rank as soon as the first one has exited, so the
    second message is not guaranteed (round 3's form of this test failed 2 runs of 3 for that reason)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["CORB_BENCH_RANK_MARK_DIR"] = str(tmp_path)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=240, env=env)
    assert p.returncode != 0
    assert sorted(os.listdir(tmp_path)) == ["rank0_of_2", "rank1_of_2"], os.listdir(tmp_path)
    assert p.stderr.count("2 ranks but") >= 1 or p.stderr.count("no MI355X visible") >= 1, p.stderr[-2000:]
    assert p.stdout.strip() == ""
