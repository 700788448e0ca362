"""Deterministic synthetic weights for the score models (host logic, numpy only).

There are no trained checkpoints or datasets in the build/bench environment
(SURVEY headline fact 5), so benchmarks and parity tests use seeded random-init
weights of the reference architecture.  The generator is PCG64-by-seed so the
weights are *regenerated* on every box instead of being committed.  Init scales
follow PyTorch's defaults for the reference modules (score_models.py:55-66,
transformer.py:12-15,66-75): U(+-1/sqrt(fan_in)) for Linear/LSTM, xavier-uniform
for in_proj, N(0,1) embedding rows (renormalised to max_norm on load), 30*N(0,1)
for the Gaussian-Fourier frequencies.  LayerNorm gains/biases and the attention
biases are perturbed away from their 1/0 defaults so that parity tests exercise
them.  Unlike the reference's deepcopy-cloned layers, every layer gets its own
draw so layer-indexing bugs are visible.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np


def _uniform(rng: np.random.Generator, shape, bound: float) -> np.ndarray:
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _common(rng, n_channels: int, max_len: int, d_model: int, with_pos: bool) -> Dict[str, np.ndarray]:
    sd: Dict[str, np.ndarray] = {}
    if with_pos:
        sd["pos_encoder.embedding.weight"] = rng.standard_normal((max_len, d_model)).astype(np.float32)
    sd["time_encoder.W"] = (rng.standard_normal((d_model + 1) // 2) * 30.0).astype(np.float32)
    sd["time_encoder.dense.weight"] = _uniform(rng, (d_model, d_model), 1 / math.sqrt(d_model))
    sd["time_encoder.dense.bias"] = _uniform(rng, (d_model,), 1 / math.sqrt(d_model))
    sd["embedder.weight"] = _uniform(rng, (d_model, n_channels), 1 / math.sqrt(n_channels))
    sd["embedder.bias"] = _uniform(rng, (d_model,), 1 / math.sqrt(n_channels))
    sd["unembedder.weight"] = _uniform(rng, (n_channels, d_model), 1 / math.sqrt(d_model))
    sd["unembedder.bias"] = _uniform(rng, (n_channels,), 1 / math.sqrt(d_model))
    return sd


def transformer_state_dict(n_channels: int, max_len: int, d_model: int = 72, num_layers: int = 10,
                           dim_feedforward: int = 2048, seed: int = 42) -> Dict[str, np.ndarray]:
    """State dict with the reference ScoreModule's parameter names."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, F = d_model, dim_feedforward
    sd = _common(rng, n_channels, max_len, d, with_pos=True)
    for i in range(num_layers):
        p = f"backbone.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = _uniform(rng, (3 * d, d), math.sqrt(6.0 / (4 * d)))
        sd[p + "self_attn.in_proj_bias"] = _uniform(rng, (3 * d,), 0.05)
        sd[p + "self_attn.out_proj.weight"] = _uniform(rng, (d, d), 1 / math.sqrt(d))
        sd[p + "self_attn.out_proj.bias"] = _uniform(rng, (d,), 0.05)
        sd[p + "linear1.weight"] = _uniform(rng, (F, d), 1 / math.sqrt(d))
        sd[p + "linear1.bias"] = _uniform(rng, (F,), 1 / math.sqrt(d))
        sd[p + "linear2.weight"] = _uniform(rng, (d, F), 1 / math.sqrt(F))
        sd[p + "linear2.bias"] = _uniform(rng, (d,), 1 / math.sqrt(F))
        sd[p + "norm1.weight"] = (1.0 + _uniform(rng, (d,), 0.1)).astype(np.float32)
        sd[p + "norm1.bias"] = _uniform(rng, (d,), 0.05)
        sd[p + "norm2.weight"] = (1.0 + _uniform(rng, (d,), 0.1)).astype(np.float32)
        sd[p + "norm2.bias"] = _uniform(rng, (d,), 0.05)
    return sd


def lstm_state_dict(n_channels: int, max_len: int, d_model: int = 72, num_layers: int = 10,
                    seed: int = 42) -> Dict[str, np.ndarray]:
    """State dict with the reference LSTMScoreModule's parameter names."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d = d_model
    sd = _common(rng, n_channels, max_len, d, with_pos=False)
    k = 1 / math.sqrt(d)
    for i in range(num_layers):
        p = f"backbone.{i}."
        sd[p + "weight_ih_l0"] = _uniform(rng, (4 * d, d), k)
        sd[p + "weight_hh_l0"] = _uniform(rng, (4 * d, d), k)
        sd[p + "bias_ih_l0"] = _uniform(rng, (4 * d,), k)
        sd[p + "bias_hh_l0"] = _uniform(rng, (4 * d,), k)
    return sd


def mlp_state_dict(n_channels: int, max_len: int, d_model: int = 72, d_mlp: int = 512, num_layers: int = 3,
                   seed: int = 42) -> Dict[str, np.ndarray]:
    """State dict with the reference MLPScoreModule's parameter names (score_models.py:392-405; the blocks are
    torchvision.ops.MLP(d, [d_mlp, d]) = Sequential(Linear, ReLU, Dropout, Linear, Dropout): indices 0 and 3)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, io = d_model, max_len * n_channels
    sd = _common(rng, n_channels, max_len, d, with_pos=False)
    sd["embedder.weight"] = _uniform(rng, (d, io), 1 / math.sqrt(io))
    sd["embedder.bias"] = _uniform(rng, (d,), 1 / math.sqrt(io))
    sd["unembedder.weight"] = _uniform(rng, (io, d), 1 / math.sqrt(d))
    sd["unembedder.bias"] = _uniform(rng, (io,), 1 / math.sqrt(d))
    for i in range(num_layers):
        p = f"backbone.{i}."
        sd[p + "0.weight"] = _uniform(rng, (d_mlp, d), 1 / math.sqrt(d))
        sd[p + "0.bias"] = _uniform(rng, (d_mlp,), 1 / math.sqrt(d))
        sd[p + "3.weight"] = _uniform(rng, (d, d_mlp), 1 / math.sqrt(d_mlp))
        sd[p + "3.bias"] = _uniform(rng, (d,), 1 / math.sqrt(d_mlp))
    return sd


def noise_stream(shape, count: int, seed: int):
    """``count`` independent N(0,1) fp32 arrays of ``shape`` (injected-noise parity runs)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(count):
        yield rng.standard_normal(shape).astype(np.float32)
