"""``fdiff.utils.fourier`` mirror: ``dft`` / ``idft`` (reference fourier.py:8-94).

Packed ortho real FFT along dim 1 of (B, L, C), computed by libffd's LDS-staged
Stockham kernel (csrc/ffd_fft.hip).  Tensors that live on the host (the reference
applies ``idft`` to the CPU tensor returned by ``sample``, cmd/sample.py:111-113) are
staged through the GPU and returned on their original device; without a GPU the call
raises -- there is no CPU implementation.
"""
from __future__ import annotations

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
