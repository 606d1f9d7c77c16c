"""One-client-per-GPU helpers over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The extract+match path shards by client with NO data-path collective: the only collectives
of bench.py are the barrier and the MAX/SUM reductions below.  The client->server map push is NOT here: it is
corb_map_push_ex in the C-ABI library (csrc/corb_comm.cpp; RCCL on device records), tested by tests/test_push_plan.py
(CPU, mock transport) and tests/test_gpu_mapstore.py (four ranks over the in-process transport on one GPU).
"""


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


def gather_scalars(dist, value, device="cpu"):
    """every rank's value, in rank order, on every rank"""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.cpu()]
