"""``cmd/benchmark_cache.py::benchmark_sampling`` mirror (reference cmd/benchmark_cache.py:42-112).

Same arguments, same sequence of calls -- ``DiffusionSampler(sample_batch_size=1, ...)``, ``cache.reset()``, a
10-step warm-up sample, ``cache.reset()``, the timed ``sample`` -- and the same result dictionary, so the rest of
the reference's script (speed-up tables, plots) runs on it unchanged.  The only addition is ``sample_batch_size``
(the reference hard-codes 1) and a device synchronisation around the timed region (the
reference's ``time.time()`` pair does not synchronise; on a GPU that would time the enqueue, not the work --
``sample`` ends with ``X.cpu()`` anyway, which synchronises).

``run_cache_benchmark`` is the body of the reference script's ``main`` (cmd/benchmark_cache.py:159-422) without the
checkpoint glob, CSV and plots: the same sequence of ``benchmark_sampling`` calls on ONE model -- no cache, default
cache, cache + FreSca, then the K / R / tau_0 / freq_decomp_interval / FreSca-high-scale grids -- returned as the
rows of the reference's result table (same column names).
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


# cmd/benchmark_cache.py:274-422: (label format, table "Parameter", cache kwarg or None for FreSca, values)
ABLATION_GRID = (
    ("K={}", "K", "K", (0, 3, 5, 10)),
    ("R={}", "R", "R", (5, 10, 20, 50)),
    ("tau_0={}", "tau_0", "tau_0", (0.05, 0.1, 0.2, 0.5)),
    ("interval={}", "freq_decomp_interval", "freq_decomp_interval", (5, 10, 20, 50)),
    ("FreSca h={}", "fresca_high_scale", None, (1.0, 1.2, 1.5, 2.0)),
)


def _row(config, parameter, value, res, t_base) -> dict:
    n, nd, t = res["num_samples"], res["num_diffusion_steps"], res["elapsed_time"]
    st = res.get("cache_stats") or {}
    return {"Config": config, "Parameter": parameter, "Value": value, "Time (s)": t, "Speedup": t_base / t,
            "Time per Sample (s)": t / n, "Time per Step (s)": t / (n * nd),
            "Cache Hit Ratio": st.get("cache_hit_ratio", 0.0), "Cache Ratio": st.get("cache_ratio", 0.0),
            "Freq Decomp Count": st.get("freq_decomp_count", 0)}


def run_cache_benchmark(score_model, num_samples: int = 10, num_diffusion_steps: int = 100, ablation: bool = True,
                        sample_batch_size: int = 1) -> list:
    common = dict(score_model=score_model, num_samples=num_samples, num_diffusion_steps=num_diffusion_steps,
                  sample_batch_size=sample_batch_size)
    fresca = dict(use_cache=True, cache_kwargs={"use_fresca_in_cache": True}, use_fresca=True)
    base = benchmark_sampling(use_cache=False, **common)  # benchmark_cache.py:161-166
    t0 = base["elapsed_time"]
    rows = [_row("No Cache", "baseline", None, base, t0)]
    rows.append(_row("E2-CRF (default)", "default", None,
                     benchmark_sampling(use_cache=True, cache_kwargs={}, use_fresca=False, **common), t0))
    rows.append(_row("E2-CRF + FreSca", "fresca", None,
                     benchmark_sampling(fresca_kwargs={"fresca_high_scale": 1.5, "fresca_cutoff_ratio": 0.5}, **fresca,
                                        **common), t0))
    if ablation:
        for label, parameter, kwarg, values in ABLATION_GRID:
            for v in values:
                if kwarg is not None:
                    res = benchmark_sampling(use_cache=True, cache_kwargs={kwarg: v}, use_fresca=False, **common)
                else:
                    res = benchmark_sampling(fresca_kwargs={"fresca_high_scale": v, "fresca_cutoff_ratio": 0.5}, **fresca,
                                             **common)
                rows.append(_row(label.format(v), parameter, v, res, t0))
    return rows
