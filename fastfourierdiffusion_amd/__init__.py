"""fastfourierdiffusion_amd -- MI355X-native sampling path of frequency-domain diffusion.

A drop-in for the *sampling* surface of NoakLiu/FastFourierDiffusion (``fdiff``): the
sub-modules mirror the reference's import paths

    fdiff.sampling.sampler.DiffusionSampler      -> .sampling.sampler
    fdiff.models.score_models.{ScoreModule,LSTMScoreModule,MLPScoreModule} -> .models.score_models
    fdiff.schedulers.sde.{SDE,VPScheduler,VEScheduler}      -> .schedulers.sde
    fdiff.utils.caching.E2CRFCache               -> .utils.caching
    fdiff.utils.fourier.{dft,idft,spectral_density,frequency_decompose_*,predict_hermite} -> .utils.fourier
    fdiff.utils.dataclasses.DiffusableBatch      -> .utils.dataclasses
    fdiff.utils.extraction.{get_best_checkpoint,get_model_type,flatten_config} -> .utils.extraction

and ``install_as_fdiff()`` registers them under the ``fdiff.*`` names so that existing
scripts and Hydra ``_target_`` strings resolve unchanged.  All arithmetic runs in
hand-written HIP kernels for gfx950 behind the C ABI in include/ffd.h (libffd.so);
PyTorch-ROCm only provides device memory, streams and parameter containers.  There is
no CPU fallback: without the built library and an MI355X the compute entry points raise.
"""
from __future__ import annotations

import importlib
import sys
import types

__version__ = "0.1.0"

_MIRROR = {
    "fdiff.sampling.sampler": "fastfourierdiffusion_amd.sampling.sampler",
    "fdiff.models.score_models": "fastfourierdiffusion_amd.models.score_models",
    "fdiff.models.transformer": "fastfourierdiffusion_amd.models.transformer",
    "fdiff.schedulers.sde": "fastfourierdiffusion_amd.schedulers.sde",
    "fdiff.utils.caching": "fastfourierdiffusion_amd.utils.caching",
    "fdiff.utils.fourier": "fastfourierdiffusion_amd.utils.fourier",
    "fdiff.utils.fresca": "fastfourierdiffusion_amd.utils.fresca",
    "fdiff.utils.dataclasses": "fastfourierdiffusion_amd.utils.dataclasses",
    "fdiff.utils.extraction": "fastfourierdiffusion_amd.utils.extraction",
}


def install_as_fdiff(force: bool = False) -> None:
    """Register this package's modules under the reference's ``fdiff.*`` names."""
    if "fdiff" in sys.modules and not force and not getattr(sys.modules["fdiff"], "__ffd_amd__", False):
        raise RuntimeError("a different `fdiff` package is already imported; pass force=True to shadow it")
    for pkg in ("fdiff", "fdiff.sampling", "fdiff.models", "fdiff.schedulers", "fdiff.utils"):
        m = types.ModuleType(pkg)
        m.__path__ = []  # mark as package
        m.__ffd_amd__ = True
        sys.modules[pkg] = m
    for alias, target in _MIRROR.items():
        mod = importlib.import_module(target)
        sys.modules[alias] = mod
        parent, _, leaf = alias.rpartition(".")
        setattr(sys.modules[parent], leaf, mod)
    for sub in ("sampling", "models", "schedulers", "utils"):
        setattr(sys.modules["fdiff"], sub, sys.modules[f"fdiff.{sub}"])


def __getattr__(name):  # lazy top-level conveniences
    table = {
        "DiffusionSampler": ("fastfourierdiffusion_amd.sampling.sampler", "DiffusionSampler"),
        "ScoreModule": ("fastfourierdiffusion_amd.models.score_models", "ScoreModule"),
        "LSTMScoreModule": ("fastfourierdiffusion_amd.models.score_models", "LSTMScoreModule"),
        "MLPScoreModule": ("fastfourierdiffusion_amd.models.score_models", "MLPScoreModule"),
        "VPScheduler": ("fastfourierdiffusion_amd.schedulers.sde", "VPScheduler"),
        "VEScheduler": ("fastfourierdiffusion_amd.schedulers.sde", "VEScheduler"),
        "E2CRFCache": ("fastfourierdiffusion_amd.utils.caching", "E2CRFCache"),
        "DiffusableBatch": ("fastfourierdiffusion_amd.utils.dataclasses", "DiffusableBatch"),
        "benchmark_sampling": ("fastfourierdiffusion_amd.benchmark", "benchmark_sampling"),
        "dft": ("fastfourierdiffusion_amd.utils.fourier", "dft"),
        "idft": ("fastfourierdiffusion_amd.utils.fourier", "idft"),
    }
    if name in table:
        mod, attr = table[name]
        return getattr(importlib.import_module(mod), attr)
    raise AttributeError(name)
