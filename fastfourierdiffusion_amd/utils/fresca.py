"""``fdiff.utils.fresca`` mirror: FreSca frequency scaling of the score
(reference src/fdiff/utils/fresca.py:111-268; 3-D (batch, seq_len, channels) case, which is
the only one the sampling path uses).

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
    if x.dim() != 3:
        if x.dim() == 4:
            raise NotImplementedError("the 2-D (batch, H, W, channels) FreSca branch is not on the sampling path")
        raise ValueError(f"Unsupported tensor dimension: {x.dim()}")
    if dim != 1:
        raise NotImplementedError("FreSca is applied along the sequence dimension (dim=1) on the sampling path")
    if cutoff_strategy not in _STRATEGY:
        raise ValueError(f"Unknown cutoff_strategy: {cutoff_strategy}")
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


def create_frequency_masks(shape, cutoff_ratio: float, cutoff_strategy: str = "spatial",
                           freq_spectrum: Optional[torch.Tensor] = None):
    """fresca.py:13-108 -- (low_pass_mask, high_pass_mask) for a 1-D (n_freq,) or 2-D (H, W) frequency grid.
    Host-side helper on a handful of elements (the sampling path computes its cutoff on the device, csrc/ffd_fft.hip);
    restated here so that code importing it from ``fdiff.utils.fresca`` keeps working."""
    device = freq_spectrum.device if freq_spectrum is not None else torch.device("cpu")
    if len(shape) == 1:
        n_freq = shape[0]
        k = torch.arange(n_freq, device=device).float()
        if cutoff_strategy == "spatial":
            low = (k <= cutoff_ratio * n_freq).float()
        elif cutoff_strategy == "energy":
            if freq_spectrum is None:
                raise ValueError("freq_spectrum required for energy-based cutoff")
            etot = torch.abs(freq_spectrum).sum()
            rc, cum = 0, 0.0
            for i in range(n_freq):
                cum += torch.abs(freq_spectrum[i]).item()
                if cum >= cutoff_ratio * etot.item():
                    rc = i
                    break
            low = (k <= rc).float()
        else:
            raise ValueError(f"Unknown cutoff_strategy: {cutoff_strategy}")
    elif len(shape) == 2:
        H, W = shape
        kx = torch.arange(H, device=device, dtype=torch.float32)
        ky = torch.arange(W, device=device, dtype=torch.float32)
        kx, ky = torch.meshgrid(kx, ky, indexing="ij")
        k_dist = torch.sqrt(kx ** 2 + ky ** 2)
        if cutoff_strategy == "spatial":
            low = (k_dist <= cutoff_ratio * min(H / 2, W / 2)).float()
        elif cutoff_strategy == "energy":
            if freq_spectrum is None:
                raise ValueError("freq_spectrum required for energy-based cutoff")
            etot = torch.abs(freq_spectrum).sum()
            rc = 0
            for R in range(int(min(H, W) / 2) + 1):
                if (torch.abs(freq_spectrum) * (k_dist <= R).float()).sum() >= cutoff_ratio * etot:
                    rc = R
                    break
            low = (k_dist <= rc).float()
        else:
            raise ValueError(f"Unknown cutoff_strategy: {cutoff_strategy}")
    else:
        raise ValueError(f"Unsupported shape dimension: {len(shape)}")
    return low, 1.0 - low


def analyze_frequency_content(x: torch.Tensor, cutoff_ratio: float = 0.5) -> dict:
    """fresca.py:271-311 -- band energies of |rfft(x)|.  The magnitudes come from the device (packed dft ->
    spectral density); the mask algebra repeats the reference's own broadcasting, including its (n_freq, 1) 2-D mask
    (only the DC bin is "low") and its shape error for batch sizes other than 1 or n_freq."""
    from .fourier import spectral_density

    mag = torch.sqrt(spectral_density(x, apply_dft=True))  # (B, n_freq, C) = |rfft(x, norm="ortho")|
    n_freq = mag.shape[1]
    low_mask, high_mask = create_frequency_masks((n_freq, 1), cutoff_ratio, "spatial")
    low_mask = low_mask.unsqueeze(0).unsqueeze(-1).to(x.device)
    high_mask = high_mask.unsqueeze(0).unsqueeze(-1).to(x.device)
    low_energy = (mag * low_mask).sum()
    high_energy = (mag * high_mask).sum()
    total_energy = mag.sum()
    return {"low_energy": low_energy, "high_energy": high_energy, "total_energy": total_energy,
            "low_energy_ratio": low_energy / (total_energy + 1e-8),
            "high_energy_ratio": high_energy / (total_energy + 1e-8)}
