"""``fdiff.schedulers.sde`` mirror: SDE / VPScheduler / VEScheduler.

Same constructor arguments, attributes (``G``, ``G_matrix``, ``timesteps``,
``step_size``, ``T``, ``eps``, ``noise_scaling``) and methods as the reference
(src/fdiff/schedulers/sde.py:13-246).  The host keeps the small tables as torch CPU
tensors exactly like the reference; the per-step arithmetic (``step``,
``prior_sampling`` on device tensors) runs in libffd's fused HIP kernel
(csrc/ffd_elem.hip) instead of the reference's dense diag(L x L) matmuls.
"""
from __future__ import annotations

import abc
import ctypes as C
import math
from collections import namedtuple
from typing import Optional

import torch

from .. import _native as N

SamplingOutput = namedtuple("SamplingOutput", ["prev_sample"])


class SDE(abc.ABC):
    """sde.py:13-87."""

    _sde_kind = -1

    def __init__(self, fourier_noise_scaling: bool = False, eps: float = 1e-5):
        super().__init__()
        self.noise_scaling = fourier_noise_scaling
        self.eps = eps
        self.G: Optional[torch.Tensor] = None
        self._G_dev = {}

    @property
    def T(self) -> float:
        return 1.0

    # -- tables (host, torch CPU: identical ops to the reference) ------------
    def set_noise_scaling(self, max_len: int) -> None:
        """sde.py:42-60."""
        G = torch.ones(max_len)
        if self.noise_scaling:
            G = 1 / (math.sqrt(2)) * G
            G[0] *= math.sqrt(2)
            if max_len % 2 == 0:
                G[max_len // 2] *= math.sqrt(2)
        self.G = G
        self.G_matrix = torch.diag(G)
        self._G_dev = {}
        assert G.shape[0] == max_len

    def set_timesteps(self, num_diffusion_steps: int) -> None:
        """sde.py:62-64."""
        self.timesteps = torch.linspace(1.0, self.eps, num_diffusion_steps)
        self.step_size = self.timesteps[0] - self.timesteps[1]

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """sde.py:66-77 (training-side helper; plain tensor arithmetic)."""
        mean, _ = self.marginal_prob(original_samples, timesteps)
        return mean + noise

    @abc.abstractmethod
    def marginal_prob(self, x: torch.Tensor, t: torch.Tensor):
        ...

    @abc.abstractmethod
    def _sde_ab(self) -> tuple:
        ...

    # -- device plumbing -------------------------------------------------------
    def _desc(self) -> N.SdeDesc:
        a, b = self._sde_ab()
        return N.SdeDesc(self._sde_kind, 0, float(a), float(b))

    def _G_on(self, device: torch.device) -> torch.Tensor:
        assert self.G is not None, "call set_noise_scaling(max_len) first (sde.py:113-114 sets it lazily only in marginal_prob)"
        key = str(device)
        g = self._G_dev.get(key)
        if g is None or g.numel() != self.G.numel():
            g = self.G.to(device=device, dtype=torch.float32).contiguous()
            self._G_dev[key] = g
        return g

    def prior_sampling(self, shape: tuple, device: Optional[torch.device] = None) -> torch.Tensor:
        """sde.py:79-87 (VE: 125-127).  Like the reference, the N(0,1) draw comes from
        torch's CPU generator (``torch.randn(*shape)``) so that seeds line up; the
        G-scaling (the reference's dense G_matrix @ z) runs in the HIP kernel.  With
        ``device=None`` the result is returned on the CPU as in the reference; passing a
        cuda device (extension) keeps it resident."""
        assert self.G is not None
        z = torch.randn(*shape)
        if not torch.cuda.is_available():
            raise N.FFDError("prior_sampling needs an MI355X (gfx950) device; there is no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        zd = z.to(dev)
        x = torch.empty_like(zd)
        B, L, Cn = shape
        desc = self._desc()
        rc = N.lib().ffd_prior(C.byref(desc), x.data_ptr(), zd.data_ptr(), self._G_on(dev).data_ptr(), 0, 0, B, L,
                               Cn, N.current_stream_ptr(dev))
        N.check(rc, None, "ffd_prior")
        return x.cpu() if device is None else x

    def _prior_scale(self) -> float:
        return 1.0

    def step(self, model_output: torch.Tensor, timestep: float, sample: torch.Tensor,
             noise: Optional[torch.Tensor] = None) -> SamplingOutput:
        """sde.py:129-165 / 215-246.  ``noise`` (extension) injects z; by default
        z = torch.randn_like(sample), exactly where the reference draws it."""
        sample = N.require_gpu_tensor(sample, "sample")
        model_output = N.require_gpu_tensor(model_output, "model_output")
        assert self.step_size > 0
        B, L, Cn = sample.shape
        z = torch.randn_like(sample) if noise is None else N.require_gpu_tensor(noise, "noise")
        x = sample.clone()
        desc = self._desc()
        rc = N.lib().ffd_sde_step(C.byref(desc), x.data_ptr(), model_output.data_ptr(),
                                  self._G_on(sample.device).data_ptr(), float(timestep), float(self.step_size),
                                  z.data_ptr(), 0, 0, 0, B, L, Cn, N.current_stream_ptr(sample.device))
        N.check(rc, None, "ffd_sde_step")
        return SamplingOutput(prev_sample=x)


class VEScheduler(SDE):
    """sde.py:90-165."""

    _sde_kind = N.FFD_SDE_VE

    def __init__(self, sigma_min: float = 0.01, sigma_max: float = 50.0, fourier_noise_scaling: bool = False,
                 eps: float = 1e-5):
        super().__init__(fourier_noise_scaling=fourier_noise_scaling, eps=eps)
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max

    def _sde_ab(self):
        return self.sigma_min, self.sigma_max

    def _prior_scale(self) -> float:
        return float(self.sigma_max)

    def marginal_prob(self, x: torch.Tensor, t: torch.Tensor):
        """sde.py:106-123 (training-side; plain tensor arithmetic)."""
        if self.G is None:
            self.set_noise_scaling(x.shape[1])
        sigma_min = torch.tensor(self.sigma_min).type_as(t)
        sigma_max = torch.tensor(self.sigma_max).type_as(t)
        std = (sigma_min * (sigma_max / sigma_min) ** t).view(-1, 1) * self.G.to(x.device)
        return x, std


class VPScheduler(SDE):
    """sde.py:168-246."""

    _sde_kind = N.FFD_SDE_VP

    def __init__(self, beta_min: float = 0.1, beta_max: float = 20.0, fourier_noise_scaling: bool = False,
                 eps: float = 1e-5):
        super().__init__(fourier_noise_scaling=fourier_noise_scaling, eps=eps)
        self.beta_0 = beta_min
        self.beta_1 = beta_max

    def _sde_ab(self):
        return self.beta_0, self.beta_1

    def get_beta(self, timestep: float) -> float:
        return self.beta_0 + timestep * (self.beta_1 - self.beta_0)

    def marginal_prob(self, x: torch.Tensor, t: torch.Tensor):
        """sde.py:186-210 (training-side; plain tensor arithmetic)."""
        if self.G is None:
            self.set_noise_scaling(x.shape[1])
        log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        mean = torch.exp(log_mean_coeff[(...,) + (None,) * len(x.shape[1:])]) * x
        std = torch.sqrt((1.0 - torch.exp(2.0 * log_mean_coeff.view(-1, 1)))) * self.G.to(x.device)
        return mean, std
