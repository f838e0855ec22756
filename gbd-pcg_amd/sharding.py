"""Batch sharding across GPUs: problems are independent, so rank g owns the
contiguous slice [g*B/G, (g+1)*B/G) and no collective touches problem data.  The only
communication is aggregation of the throughput numbers (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations


def shard_range(batch: int, rank: int, world: int):
    """Contiguous slice of `batch` problems owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def aggregate(elapsed_s: float, units: float, device=None):
    """(max elapsed over ranks, total units over ranks).  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed_s, units
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
