"""Token sharding and timing helpers for one-process-per-GPU runs.

The RQAE hot path has no cross-token operation (every reduction in rqae/model.py:199-252 is
over the last dim), so N tokens shard as contiguous ranges, weights are replicated, and no
collective touches the data path.  The only communication is the timing reduction below.
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Tuple


def shard_range(n_tokens: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous token range [start, end) of `rank` out of `world`; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_tokens, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def timed_steps(step: Callable[[], None], steps: int, warmup: int,
                barrier: Callable[[], None]) -> float:
    """`warmup` untimed steps, then exactly `steps` timed ones bracketed by `barrier()`
    (which must also drain the device) on both sides.  Returns seconds on this rank."""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """MAX all-reduce of a scalar over the default process group (identity without one)."""
    if dist is None or not dist.is_initialized():
        return seconds
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(tokens_per_rank: int, world: int, steps: int, max_seconds: float) -> float:
    """Whole-job tokens/s: every rank's tokens over the slowest rank's time."""
    return tokens_per_rank * world * steps / max_seconds
