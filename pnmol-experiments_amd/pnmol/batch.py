"""Batches of independent PDE problems, one problem per GPU / rank.

The time recursion of one problem is strictly sequential and its state is one dense covariance, so the
path shards only across problems: rank g solves problem g with its own device context and there is no
collective in the data path.  The only communication is the final gather of the per-problem read-outs
(means, stds, diffusions), done with `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU node,
"gloo" in the CPU tests).
"""

import numpy as np


def diffusion_sweep(index, count=8, lo=0.01, hi=0.1):
    """kappa_g = lo * (hi/lo)^(g/(count-1)): the parameter sweep of BASELINE.json's 8-problem batch."""
    if count == 1:
        return 0.05
    return float(lo * (hi / lo) ** (index / (count - 1)))


def shard(num_problems, rank, world_size):
    """Indices of the problems rank `rank` owns (round-robin; one per rank when num_problems == world_size)."""
    return list(range(rank, num_problems, world_size))


def gather_readouts(payload, dist=None, device="cpu", force=False):
    """All-gather equally shaped float64 arrays: returns an array (world_size, *payload.shape).

    A one-rank group skips the collective unless `force` (used to exercise the RCCL path on a one-GPU box)."""
    payload = np.ascontiguousarray(payload, dtype=np.float64)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return payload[None]
    import torch

    t = torch.from_numpy(payload).to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


def max_over_ranks(value, dist=None, device="cpu", force=False):
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
