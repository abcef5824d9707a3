"""Frame-level sharding across the GPUs of one node: frames are independent once captured, so rank r takes frames
r, r+world, ... and the only exchange is the gather of the tiny per-frame results (state increment, n_eff, ...).  No data-path
collective.  Works with any torch.distributed backend ("nccl" == RCCL on the GPUs, "gloo" in the CPU tests)."""
import numpy as np


def frames_for_rank(n_frames, rank, world):
    return list(range(rank, n_frames, world))


def gather_results(local, n_frames, dist=None, device="cpu"):
    """local: float64 [n_local, K] results of frames_for_rank(n_frames, rank, world) in that order.
    Returns float64 [n_frames, K] in frame order on every rank (one all_gather of a padded buffer)."""
    local = np.ascontiguousarray(local, np.float64).reshape(len(local), -1)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert len(local) == n_frames
        return local
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    K = local.shape[1] if local.size else 0
    kk = torch.tensor([K], dtype=torch.int64, device=device)
    dist.all_reduce(kk, op=dist.ReduceOp.MAX)
    K = int(kk.item())
    per = (n_frames + world - 1) // world
    buf = torch.zeros((per, K), dtype=torch.float64, device=device)
    if len(local):
        buf[: len(local)] = torch.from_numpy(local).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = np.zeros((n_frames, K))
    for r in range(world):
        idx = frames_for_rank(n_frames, r, world)
        res[idx] = out[r][: len(idx)].cpu().numpy()
    return res


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX of a python float over all ranks (the bench's timing rule)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
