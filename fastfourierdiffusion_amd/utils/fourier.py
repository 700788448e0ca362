"""``fdiff.utils.fourier`` mirror: ``dft`` / ``idft`` (reference fourier.py:8-94), ``spectral_density``
(:97-131) and the FreqCa helpers ``frequency_decompose_fft`` / ``_dct`` (:219-305) and ``predict_hermite``
(:397-497).

Packed ortho real FFT along dim 1 of (B, L, C), computed by libffd's LDS-staged
Stockham kernel (csrc/ffd_fft.hip).  Tensors that live on the host (the reference
applies ``idft`` to the CPU tensor returned by ``sample``, cmd/sample.py:111-113) are
staged through the GPU and returned on their original device; without a GPU the call
raises -- there is no CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from .. import _native as N


def _run(x: torch.Tensor, inverse: bool) -> torch.Tensor:
    if torch.is_complex(x):  # fourier.py:18-19
        x = torch.real(x)
    assert x.dim() == 3, f"expected (batch_size, max_len, n_channels), got {tuple(x.shape)}"
    src_device = x.device
    if src_device.type != "cuda":
        if not torch.cuda.is_available():
            raise N.FFDError("dft/idft need an MI355X (gfx950) device; there is no CPU fallback")
        x = x.to("cuda")
    xd = x.detach().to(torch.float32).contiguous()
    out = torch.empty_like(xd)
    B, L, Cn = xd.shape
    fn = N.lib().ffd_idft if inverse else N.lib().ffd_dft
    rc = fn(xd.data_ptr(), out.data_ptr(), B, L, Cn, N.current_stream_ptr(xd.device))
    N.check(rc, None, "ffd_idft" if inverse else "ffd_dft")
    return out.to(src_device) if src_device.type != "cuda" else out


def dft(x: torch.Tensor) -> torch.Tensor:
    """fourier.py:8-52."""
    return _run(x, inverse=False)


def idft(x: torch.Tensor) -> torch.Tensor:
    """fourier.py:55-94."""
    return _run(x, inverse=True)


def _run_affine(x: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, inverse: bool) -> torch.Tensor:
    assert x.dim() == 3, f"expected (batch_size, max_len, n_channels), got {tuple(x.shape)}"
    src_device = x.device
    xd = _on_gpu(x, "dft/idft")
    B, L, Cn = xd.shape
    mu = torch.broadcast_to(torch.as_tensor(mean, dtype=torch.float32), (L, Cn)).to(xd.device).contiguous()
    sd = torch.broadcast_to(torch.as_tensor(std, dtype=torch.float32), (L, Cn)).to(xd.device).contiguous()
    out = torch.empty_like(xd)
    fn = N.lib().ffd_unstandardize_idft if inverse else N.lib().ffd_dft_standardize
    N.check(fn(xd.data_ptr(), out.data_ptr(), mu.data_ptr(), sd.data_ptr(), B, L, Cn, N.current_stream_ptr(xd.device)),
            None, "ffd_unstandardize_idft" if inverse else "ffd_dft_standardize")
    return out.to(src_device) if src_device.type != "cuda" else out


def unstandardize_idft(x: torch.Tensor, feature_mean: torch.Tensor, feature_std: torch.Tensor) -> torch.Tensor:
    """The post-step of ``cmd/sample.py:107-113`` in one kernel: ``idft(X * feature_std + feature_mean)``
    (samples drawn in the standardised frequency domain -> time series on the data's scale)."""
    return _run_affine(x, feature_mean, feature_std, inverse=True)


def dft_standardize(x: torch.Tensor, feature_mean: torch.Tensor, feature_std: torch.Tensor) -> torch.Tensor:
    """The ingest twin, ``DiffusionDataset`` (datamodules.py:42-43,61-62): ``(dft(X) - feature_mean) / feature_std``."""
    return _run_affine(x, feature_mean, feature_std, inverse=False)


def _on_gpu(x: torch.Tensor, what: str) -> torch.Tensor:
    if x.device.type != "cuda":
        if not torch.cuda.is_available():
            raise N.FFDError(f"{what} needs an MI355X (gfx950) device; there is no CPU fallback")
        x = x.to("cuda")
    return x.detach().to(torch.float32).contiguous()


def spectral_density(x: torch.Tensor, apply_dft: bool = True) -> torch.Tensor:
    """fourier.py:97-131: |X_k|^2 per frequency, (B, L, C) -> (B, ceil((L+1)/2), C)."""
    assert x.dim() == 3
    src_device = x.device
    xd = _on_gpu(x, "spectral_density")
    B, L, Cn = xd.shape
    stream = N.current_stream_ptr(xd.device)
    if apply_dft:
        xf = torch.empty_like(xd)
        N.check(N.lib().ffd_dft(xd.data_ptr(), xf.data_ptr(), B, L, Cn, stream), None, "ffd_dft")
    else:
        xf = xd
    out = torch.empty((B, math.ceil((L + 1) / 2), Cn), device=xd.device, dtype=torch.float32)
    N.check(N.lib().ffd_spectral_density(xf.data_ptr(), out.data_ptr(), B, L, Cn, stream), None, "ffd_spectral_density")
    return out.to(src_device) if src_device.type != "cuda" else out


def frequency_decompose_fft(x: torch.Tensor, low_freq_ratio: float = 0.3):
    """fourier.py:219-286: (low, high) parts of x (B, L, D) or (L, D) along the sequence axis."""
    was_2d = x.dim() == 2
    if was_2d:
        x = x.unsqueeze(0)
    assert x.dim() == 3
    src_device = x.device
    xd = _on_gpu(x, "frequency_decompose_fft")
    B, L, D = xd.shape
    low, high = torch.empty_like(xd), torch.empty_like(xd)
    rc = N.lib().ffd_freq_decompose(xd.data_ptr(), low.data_ptr(), high.data_ptr(), B, L, D, float(low_freq_ratio),
                                    N.current_stream_ptr(xd.device))
    N.check(rc, None, "ffd_freq_decompose")
    if src_device.type != "cuda":
        low, high = low.to(src_device), high.to(src_device)
    if was_2d:
        low, high = low.squeeze(0), high.squeeze(0)
    return low, high


def frequency_decompose_dct(x: torch.Tensor, low_freq_ratio: float = 0.3):
    """fourier.py:288-305: the reference returns the FFT decomposition here (the DCT body is unreachable)."""
    return frequency_decompose_fft(x, low_freq_ratio)


def predict_hermite(history, timesteps, target_timestep: float, order: int = 2) -> torch.Tensor:
    """fourier.py:397-497: Hermite least-squares extrapolation of a list of equally shaped tensors."""
    assert len(history) >= 1 and len(history) == len(timesteps)
    src_device = history[0].device
    stack = _on_gpu(torch.stack(list(history), dim=0), "predict_hermite")
    K = stack.shape[0]
    out = torch.empty_like(stack[0])
    ts = (C.c_double * K)(*[float(t) for t in timesteps])
    rc = N.lib().ffd_hermite_predict(stack.data_ptr(), ts, float(target_timestep), int(order), out.data_ptr(), K,
                                     out.numel(), N.current_stream_ptr(stack.device))
    N.check(rc, None, "ffd_hermite_predict")
    return out.to(src_device) if src_device.type != "cuda" else out
