"""Batch sharding for multi-GPU sampling (host logic only).

Sampling is embarrassingly parallel over the batch axis (no cross-sample operator in
ScoreModule.forward, the SDE step or idft), so an N-GPU node is N independent shards:
weights replicated (12.8 MB), contiguous sample ranges per rank, Philox noise keyed by
*global* sample index so the noise does not depend on N (the samples then agree to fp32 rounding,
not bit for bit: the shard size selects the kernels, see sampling/sampler.py).  There is no data-path collective;
torch.distributed (RCCL on GPUs, gloo in CPU tests) only carries the barrier and the
max-over-ranks of elapsed time.  With the E2-CRF cache each shard behaves as an
independent reference run (its own step-0 table from its own element 0); the same
holds for FreSca's ``energy`` cutoff, a mean over the local batch (fresca.py:150-158): both are
per-batch statistics in the reference, so a shard reproduces the reference run with
``sample_batch_size`` = shard size, not a slice of a larger batch.
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_range(num_samples: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(offset, count) of `rank`'s contiguous sample range; counts differ by at most 1."""
    assert world_size >= 1 and 0 <= rank < world_size and num_samples >= 0
    base, rem = divmod(num_samples, world_size)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def reduce_max_seconds(seconds: float, device: Optional[object]) -> float:
    """MAX over ranks of a wall-clock interval (identity when not distributed)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_seconds(seconds: float, device: Optional[object]) -> list:
    """Every rank's wall-clock interval, in rank order (a one-element list when not distributed)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [float(seconds)]
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]
