"""``cmd/benchmark_cache.py::benchmark_sampling`` mirror (reference cmd/benchmark_cache.py:42-112).

Same arguments, same sequence of calls -- ``DiffusionSampler(sample_batch_size=1, ...)``, ``cache.reset()``, a
10-step warm-up sample, ``cache.reset()``, the timed ``sample`` -- and the same result dictionary, so the rest of
the reference's script (speed-up tables, plots) runs on it unchanged.  The only addition is ``sample_batch_size``
(the reference hard-codes 1) and a device synchronisation around the timed region (the
reference's ``time.time()`` pair does not synchronise; on a GPU that would time the enqueue, not the work --
``sample`` ends with ``X.cpu()`` anyway, which synchronises).
"""
from __future__ import annotations

import time
from typing import Optional

import torch

from .sampling.sampler import DiffusionSampler


def benchmark_sampling(score_model, num_samples: int = 10, num_diffusion_steps: int = 100, use_cache: bool = False,
                       cache_kwargs: Optional[dict] = None, use_fresca: bool = False,
                       fresca_kwargs: Optional[dict] = None, sample_batch_size: int = 1) -> dict:
    fresca_kwargs = dict(fresca_kwargs or {})
    if use_fresca:  # benchmark_cache.py:63-68
        fresca_kwargs.setdefault("fresca_low_scale", 1.0)
        fresca_kwargs.setdefault("fresca_high_scale", 1.5)
        fresca_kwargs.setdefault("fresca_cutoff_ratio", 0.5)
        fresca_kwargs.setdefault("fresca_cutoff_strategy", "energy")
    sampler = DiffusionSampler(score_model=score_model, sample_batch_size=sample_batch_size, use_cache=use_cache,
                               cache_kwargs=cache_kwargs, use_fresca=use_fresca, **fresca_kwargs)
    if use_cache and score_model.cache is not None:
        score_model.cache.reset()
    _ = sampler.sample(num_samples=1, num_diffusion_steps=10)  # warm-up, benchmark_cache.py:85
    if use_cache and score_model.cache is not None:
        score_model.cache.reset()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    start_time = time.time()
    samples = sampler.sample(num_samples=num_samples, num_diffusion_steps=num_diffusion_steps)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    elapsed_time = time.time() - start_time
    cache_stats = {}
    if use_cache and score_model.cache is not None:
        cache_stats = score_model.cache.get_cache_stats()
    return {"elapsed_time": elapsed_time, "samples": samples, "cache_stats": cache_stats, "num_samples": num_samples,
            "num_diffusion_steps": num_diffusion_steps}
