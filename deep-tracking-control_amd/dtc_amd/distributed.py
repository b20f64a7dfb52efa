"""Data-parallel glue: one process per GPU, envs sharded across ranks, RCCL collectives
through torch.distributed ("nccl" IS RCCL on ROCm; "gloo" in the CPU tests).  These helpers are the ONLY
places where the hot path talks to other ranks:

  * `allreduce_mean_(flat_grad)`       two gradient buckets per optimiser step (7.42 MB VAE step / 7.76 MB policy step in total);
                                       the policy step's first bucket carries a 4-float header in front of the gradients
                                       whose slot 0 is the KL statistic, so that every rank takes the same learning-rate
                                       branch (ppo.py:301-307) without a collective of its own;
  * `allreduce_sum_(stat)`             the two advantage-normalisation scalars (sum, sum of squared deviations).

CE-net outlier statistics and the mini-batch permutation stay rank-local.  Everything else (planner,
GAE scan, forward, backward, Adam) is embarrassingly parallel over envs; parameters stay bit-identical
across ranks because every rank applies the same averaged gradient with the same learning rate.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# ---- collective trace (DTC_DP_TRACE=1 or `trace_collectives(True)`): every collective the hot path issues is recorded as
# (op, numel, dtype, stream) in issue order.  All ranks must issue the SAME sequence -- RCCL matches collectives of one
# communicator by order, so a rank that issues its side-stream bucket exchange and its main-stream KL all-reduce in
# another order than its peers deadlocks (gloo would not show it).  `assert_same_collective_sequence()` all-gathers
# the logs and compares them; tests and the N > 1 rehearsal of bench.py call it.
_TRACE = os.environ.get("DTC_DP_TRACE", "0") == "1"
_LOG: list = []


def trace_collectives(on: bool = True):
    global _TRACE
    _TRACE = bool(on)
    _LOG.clear()


def collective_log():
    return list(_LOG)


def _record(op: str, t: torch.Tensor):
    if _TRACE:
        stream = "cpu"
        if t.is_cuda:
            cur = torch.cuda.current_stream(t.device)
            stream = "default" if cur == torch.cuda.default_stream(t.device) else "side"
        _LOG.append((op, int(t.numel()), str(t.dtype).replace("torch.", ""), stream))


def assert_same_collective_sequence():
    """Every rank issued the same collectives in the same order (names, sizes, dtypes, stream kind); returns the
    sequence.  Itself a collective (all_gather_object): call it at the same point on every rank."""
    mine = collective_log()
    if world_size() == 1:
        return mine
    logs = [None] * world_size()
    was = _TRACE
    dist.all_gather_object(logs, mine)
    for r, other in enumerate(logs):
        if other != mine:
            n = next((i for i, (a, b) in enumerate(zip(mine, other)) if a != b), min(len(mine), len(other)))
            raise AssertionError(f"rank {rank()} and rank {r} issued different collective sequences "
                                 f"({len(mine)} vs {len(other)} entries); first difference at #{n}: "
                                 f"{mine[n] if n < len(mine) else None} vs {other[n] if n < len(other) else None}")
    assert was == _TRACE
    return mine


def bytes_reduced(log=None) -> int:
    """Payload bytes of the all-reduces in a collective log (what one rank contributes per recorded span)."""
    size = dict(float32=4, float64=8, int64=8, int32=4)
    return sum(n * size.get(dt, 4) for op, n, dt, _ in (collective_log() if log is None else log) if op.startswith("all_reduce"))


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def backend() -> str:
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else ""


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        _record("all_reduce_sum", t)
        dist.all_reduce(t)
    return t


def allreduce_mean_(t: torch.Tensor) -> torch.Tensor:
    """In-place mean over the ranks.  RCCL averages inside the collective (ReduceOp.AVG: no second pass over the
    bucket); gloo has no AVG, there the sum is scaled afterwards (CPU tests, one-GPU rehearsals)."""
    w = world_size()
    if w > 1:
        _record("all_reduce_mean", t)
        if dist.get_backend() == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
        else:
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
        _record("broadcast", flat)
        dist.broadcast(flat, src)
    return flat
