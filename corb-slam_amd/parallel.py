"""One-client-per-GPU helpers over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The extract+match path shards by client with NO data-path collective: the only collectives
of bench.py are the barrier and the MAX/SUM reduction below.  `gather_keyframes` is the client->server map
push (replaces the 6-second ROS service batch of corbslam_client/src/DataDriver.cc:135-193 when clients run
one per GPU): a padded gather of keyframe SoA blocks to the server rank.
"""
import numpy as np


def client_frame_offset(rank, frames_per_client=64):
    """Each rank is one client with its own stream of synthetic frames."""
    return frames_per_client * rank


def reduce_step_time(dist, seconds, units, device="cpu"):
    """whole-job timing: MAX of the per-rank wall time, SUM of the per-rank processed units."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds), float(units)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_keyframes(dist, kp_bytes, desc, u_right, dst=0, device="cpu"):
    """Gather one keyframe block per rank on `dst`.  kp_bytes: uint8 [n,28] (cv::KeyPoint records),
    desc: uint8 [n,32], u_right: float32 [n].  Returns a list of (kp_bytes, desc, u_right) on dst, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    n = int(len(desc))
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    sizes[rank] = n
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
    nmax = int(sizes.max().item())
    block = torch.zeros((nmax, 28 + 32 + 4), dtype=torch.uint8, device=device)
    if n:
        block[:n, :28] = torch.from_numpy(np.ascontiguousarray(kp_bytes, np.uint8).reshape(n, 28)).to(device)
        block[:n, 28:60] = torch.from_numpy(np.ascontiguousarray(desc, np.uint8)).to(device)
        block[:n, 60:64] = torch.from_numpy(np.ascontiguousarray(u_right, np.float32).view(np.uint8).reshape(n, 4)).to(device)
    out = [torch.zeros_like(block) for _ in range(world)] if rank == dst else None
    dist.gather(block, out, dst=dst)
    if rank != dst:
        return None
    res = []
    for r in range(world):
        m = int(sizes[r].item()); b = out[r][:m].cpu().numpy()
        res.append((b[:, :28].copy(), b[:, 28:60].copy(), b[:, 60:64].copy().view(np.float32).reshape(m)))
    return res
