"""``fdiff.models.transformer`` mirror: the positional and time encoders
(reference src/fdiff/models/transformer.py:8-29, 61-91).

The modules own their parameters under the reference's state_dict keys and default
initialisation (so reference checkpoints load and equal seeds give equal weights).  Inside
``ScoreModule.forward`` their arithmetic is fused into libffd's embed kernel; called on
their own (as the reference's tests/test_transformer.py does) they run the standalone
entry points ``ffd_positional_encoding`` / ``ffd_time_encoding``.  No CPU fallback.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import _native as N


class PositionalEncoding(nn.Module):
    """transformer.py:8-29: learned nn.Embedding(max_len, d_model, max_norm=sqrt(d_model))."""

    def __init__(self, d_model: int, max_len: int):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings=max_len, embedding_dim=d_model, max_norm=math.sqrt(d_model))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = N.require_gpu_tensor(x, "x")
        B, L, D = x.shape
        w = self.embedding.weight
        assert w.device == x.device and L <= w.shape[0] and D == w.shape[1]
        out = torch.empty_like(x)
        # like nn.Embedding(max_norm), the looked-up rows are renormalised in place (no autograd involved)
        rc = N.lib().ffd_positional_encoding(x.data_ptr(), w.data.data_ptr(), out.data_ptr(), B, L, D,
                                             float(self.embedding.max_norm), N.current_stream_ptr(x.device))
        N.check(rc, None, "ffd_positional_encoding")
        return out


class GaussianFourierProjection(nn.Module):
    """transformer.py:61-91: fixed Gaussian frequencies W (scale 30) + Linear(d, d)."""

    def __init__(self, d_model: int, scale: float = 30.0):
        super().__init__()
        self.d_model = d_model
        self.W = nn.Parameter(torch.randn((d_model + 1) // 2) * scale, requires_grad=False)
        self.dense = nn.Linear(d_model, d_model)

    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, use_time_axis: bool = True) -> torch.Tensor:
        x = N.require_gpu_tensor(x, "x")
        t = N.require_gpu_tensor(timesteps.to(torch.float32), "timesteps")
        if use_time_axis:
            B, L, D = x.shape
        else:
            (B, D), L = x.shape, 1
        assert D == self.d_model and t.shape[0] == B
        out = torch.empty_like(x)
        work = torch.empty(B * D, device=x.device, dtype=torch.float32)
        rc = N.lib().ffd_time_encoding(x.data_ptr(), t.data_ptr(), self.W.data_ptr(), self.dense.weight.data_ptr(),
                                       self.dense.bias.data_ptr(), work.data_ptr(), out.data_ptr(), B, L, D,
                                       N.current_stream_ptr(x.device))
        N.check(rc, None, "ffd_time_encoding")
        return out
