"""``fdiff.models.transformer`` mirror: parameter containers for the positional and
time encoders (reference src/fdiff/models/transformer.py:8-29, 61-91).

They own the parameters under the reference's state_dict keys and default
initialisation (so reference checkpoints load and equal seeds give equal weights); the
arithmetic itself is fused into libffd's embed kernel (csrc/ffd_elem.hip: k_embed,
k_time_embed) and reached through ``ScoreModule.forward``.  Calling the containers
directly on cuda tensors evaluates the same kernels through a one-layer context-free
path is not provided; use ``ScoreModule``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class PositionalEncoding(nn.Module):
    """transformer.py:8-29: learned nn.Embedding(max_len, d_model, max_norm=sqrt(d_model))."""

    def __init__(self, d_model: int, max_len: int):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings=max_len, embedding_dim=d_model, max_norm=math.sqrt(d_model))

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover - fused into ScoreModule.forward
        raise NotImplementedError(
            "PositionalEncoding is fused into ScoreModule.forward (libffd k_embed); it has no standalone kernel")


class GaussianFourierProjection(nn.Module):
    """transformer.py:61-91: fixed Gaussian frequencies W (scale 30) + Linear(d, d)."""

    def __init__(self, d_model: int, scale: float = 30.0):
        super().__init__()
        self.d_model = d_model
        self.W = nn.Parameter(torch.randn((d_model + 1) // 2) * scale, requires_grad=False)
        self.dense = nn.Linear(d_model, d_model)

    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, use_time_axis: bool = True):  # pragma: no cover
        raise NotImplementedError(
            "GaussianFourierProjection is fused into ScoreModule.forward (libffd k_time_embed + k_embed)")
