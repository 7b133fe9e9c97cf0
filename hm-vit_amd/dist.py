"""Scene sharding across ranks (one process per GPU).  The fusion path has no data-path exchange:
scenes are independent units (batch_size 1 per GPU in the reference, train_camera.py:62-71), so the
only collectives are the timing barrier and a MAX reduction of the wall time (RCCL on GPUs, gloo in
the CPU tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_scenes(n_scenes: int, rank: int, world: int):
    """Round-robin scene ids of this rank (DistributedSampler-style, no padding)."""
    return list(range(rank, n_scenes, world))


def max_over_ranks(seconds: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_throughput(units_this_rank: int, seconds: float, device=None) -> float:
    """Whole-job units/s = units processed by ALL ranks (summed: round-robin shards are unequal when the scene count is not
    a multiple of the world size) / the slowest rank's time."""
    return sum_over_ranks(float(units_this_rank), device) / max_over_ranks(seconds, device)
