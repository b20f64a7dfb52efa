"""Data-parallel glue: one process per GPU, envs sharded across ranks, RCCL collectives
through torch.distributed ("nccl" IS RCCL on ROCm; "gloo" in the CPU tests).  These helpers are the ONLY
places where the hot path talks to other ranks:

  * `allreduce_mean_(flat_grad)`       one flat gradient bucket per optimiser step (7.4 MB / 11.5 MB);
  * `allreduce_sum_(stat)`             the two advantage-normalisation scalars (sum, sum of squared deviations);
  * `allreduce_mean_(kl)`              so that every rank takes the same learning-rate branch (ppo.py:301-307).

CE-net outlier statistics and the mini-batch permutation stay rank-local.  Everything else (planner,
GAE scan, forward, backward, Adam) is embarrassingly parallel over envs; parameters stay bit-identical
across ranks because every rank applies the same averaged gradient with the same learning rate.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        dist.all_reduce(t)
    return t


def allreduce_mean_(t: torch.Tensor) -> torch.Tensor:
    w = world_size()
    if w > 1:
        dist.all_reduce(t)
        t.mul_(1.0 / w)
    return t


def shard_range(num_items: int, r: int | None = None, w: int | None = None):
    """Contiguous env shard [lo, hi) of rank r (rank r owns envs [r*N/K, (r+1)*N/K))."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per = num_items // w
    return r * per, (r + 1) * per


def broadcast_parameters_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Make the flat parameter arena identical on all ranks (called once after construction)."""
    if world_size() > 1:
        dist.broadcast(flat, src)
    return flat
