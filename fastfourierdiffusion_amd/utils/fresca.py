"""``fdiff.utils.fresca`` mirror: FreSca frequency scaling of the score
(reference src/fdiff/utils/fresca.py:111-268): the 3-D (batch, seq_len, channels) case, which is
the only one the sampling path uses, and (round 4) the 4-D (batch, H, W, channels) case
(fresca.py:184-213; ``ffd_fresca2d``: H, W <= 256, H*W <= 4096).

``frequency_scale`` / ``apply_fresca_to_score`` keep the reference signatures; the
rFFT -> mask/scale -> irFFT sequence runs in libffd's fused LDS kernels
(csrc/ffd_fft.hip: k_fresca_*), including the batch-wide energy-cutoff reduction, which
the reference evaluates with a Python loop and ``.item()`` syncs (fresca.py:52-58).
"""
from __future__ import annotations

from typing import Literal, Optional

import torch

from .. import _native as N

_STRATEGY = {"spatial": 0, "energy": 1}


def frequency_scale(x: torch.Tensor, low_scale: float = 1.0, high_scale: float = 1.0, cutoff_ratio: float = 0.5,
                    cutoff_strategy: Literal["spatial", "energy"] = "spatial", dim: int = 1) -> torch.Tensor:
    """fresca.py:111-217."""
    if low_scale == 1.0 and high_scale == 1.0:
        return x  # fresca.py:137-138
    if x.dim() not in (3, 4):
        raise ValueError(f"Unsupported tensor dimension: {x.dim()}")
    if cutoff_strategy not in _STRATEGY:
        raise ValueError(f"Unknown cutoff_strategy: {cutoff_strategy}")
    if x.dim() == 4:  # fresca.py:184-213 (rfft2 / irfft2 over dims (1, 2) whatever `dim` says)
        xd = N.require_gpu_tensor(x, "x")
        B, H, W, Cn = xd.shape
        out = torch.empty_like(xd)
        work = torch.empty(3 * B * Cn * H * (W // 2 + 1) + 4, device=xd.device, dtype=torch.float32)
        rc = N.lib().ffd_fresca2d(xd.data_ptr(), out.data_ptr(), work.data_ptr(), B, H, W, Cn, float(low_scale),
                                  float(high_scale), float(cutoff_ratio), _STRATEGY[cutoff_strategy],
                                  N.current_stream_ptr(xd.device))
        if rc == -2:  # FFD_ERR_UNSUPPORTED
            raise NotImplementedError(f"4-D FreSca: H, W <= 256 and H*W <= 4096 (got H={H}, W={W})")
        N.check(rc, None, "ffd_fresca2d")
        return out
    if dim != 1:
        raise NotImplementedError("FreSca is applied along the sequence dimension (dim=1) on the sampling path")
    xd = N.require_gpu_tensor(x, "x")
    B, L, Cn = xd.shape
    out = torch.empty_like(xd)
    work = torch.empty(B * Cn * (L // 2 + 1) + 4, device=xd.device, dtype=torch.float32)
    rc = N.lib().ffd_fresca(xd.data_ptr(), out.data_ptr(), work.data_ptr(), B, L, Cn, float(low_scale),
                            float(high_scale), float(cutoff_ratio), _STRATEGY[cutoff_strategy],
                            N.current_stream_ptr(xd.device))
    N.check(rc, None, "ffd_fresca")
    return out


def dynamic_high_scale(high_scale: float, timestep: Optional[float], num_steps: Optional[int]) -> float:
    """fresca.py:247-257: h(t) = (1 - t/num_steps)*(h - 1) + 1 for h > 1."""
    if timestep is not None and num_steps is not None:
        t_normalized = timestep / num_steps if num_steps > 0 else 0.0
        if high_scale > 1.0:
            return (1.0 - t_normalized) * (high_scale - 1.0) + 1.0
    return high_scale


def apply_fresca_to_score(score: torch.Tensor, low_scale: float = 1.0, high_scale: float = 1.0,
                          cutoff_ratio: float = 0.5, cutoff_strategy: Literal["spatial", "energy"] = "energy",
                          timestep: Optional[float] = None, num_steps: Optional[int] = None) -> torch.Tensor:
    """fresca.py:220-268."""
    return frequency_scale(score, low_scale=low_scale,
                           high_scale=dynamic_high_scale(high_scale, timestep, num_steps),
                           cutoff_ratio=cutoff_ratio, cutoff_strategy=cutoff_strategy)
