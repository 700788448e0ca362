"""``fdiff.utils.caching`` mirror: ``E2CRFCache`` (reference caching.py:19-653).

Host-side state machine only: the gate (a pure function of the step index), the
counters and the CRF slot.  The K/V tables (NL, H, L, hd) themselves live in the
native context's HBM (csrc/ffd_api.hip) and are read by the attention kernel; the
cache object that a model's layers were *first* bound to is the one whose ``reset``
and statistics reach them (reference quirk Q5, score_models.py:232).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import _native as N


class E2CRFCache:
    def __init__(self, num_layers: int, max_len: int, device: torch.device, K: int = 5, R: int = 10,
                 tau_0: float = 0.1, tau_warn: float = 0.5, use_freqca: bool = False, freq_decomp: str = "dct",
                 low_freq_ratio: float = 0.3, max_history: int = 10, hermite_order: int = 3,
                 freq_decomp_interval: int = 10, use_fresca_in_cache: bool = False,
                 fresca_adaptive_threshold: bool = False):
        # caching.py:28-113 -- same kwargs (an unknown kwarg such as README's
        # `random_probe_ratio` raises TypeError here exactly as in the reference, Q6)
        self.num_layers = num_layers
        self.max_len = max_len
        self.device = device
        self.K = K
        self.R = R
        self.tau_0 = tau_0
        self.tau_warn = tau_warn
        self.use_freqca = use_freqca
        self.freq_decomp = freq_decomp
        self.low_freq_ratio = low_freq_ratio
        self.max_history = max_history
        self.hermite_order = hermite_order
        self.freq_decomp_interval = freq_decomp_interval
        self.use_fresca_in_cache = use_fresca_in_cache
        self.fresca_adaptive_threshold = fresca_adaptive_threshold
        self.crf_cache: Optional[torch.Tensor] = None
        # FreqCa state (caching.py:103-106)
        self.crf_low_cache: Optional[torch.Tensor] = None
        self.crf_high_history: list = []
        self.crf_timestep_history: list = []
        self.stats = {"recompute_count": 0, "cache_hit_count": 0}
        self.current_step = 0
        self._bound_model = None  # the score model whose native tables this object controls

    # caching.py:115-129
    def reset(self) -> None:
        self.crf_cache = None
        self.crf_low_cache = None
        self.crf_high_history = []
        self.crf_timestep_history = []
        self.stats = {"recompute_count": 0, "cache_hit_count": 0}
        self.current_step = 0
        if self._bound_model is not None:
            self._bound_model._native_cache_reset()

    # caching.py:131-181
    def determine_recompute_set(self, x_tilde, event_intensity: float, step: int) -> set:
        n = N.lib().ffd_host_gate(int(step), int(self.max_len), int(self.K), int(self.R))
        return set(range(n))

    # caching.py:459-522
    def update_crf(self, crf: torch.Tensor, timestep: Optional[float] = None) -> None:
        needs_crf = self.use_freqca or (self.current_step % self.R == 0)
        if needs_crf:
            self.crf_cache = crf.detach()
        if self.use_freqca:
            should_decomp = self.current_step % self.freq_decomp_interval == 0 or self.current_step == 0
            if should_decomp and needs_crf:
                from .fourier import frequency_decompose_dct, frequency_decompose_fft

                fn = frequency_decompose_fft if self.freq_decomp == "fft" else frequency_decompose_dct
                crf_low, crf_high = fn(crf, self.low_freq_ratio)
                self._push_decomposition(crf_low, crf_high, timestep)

    def _push_decomposition(self, crf_low: torch.Tensor, crf_high: torch.Tensor, timestep: Optional[float]) -> None:
        """caching.py:506-522: keep the low part, append the high part to the bounded history."""
        self.crf_low_cache = crf_low.detach()
        self.crf_high_history.append(crf_high.detach())
        if timestep is not None:
            self.crf_timestep_history.append(timestep)
        if len(self.crf_high_history) > self.max_history:
            self.crf_high_history.pop(0)
            if self.crf_timestep_history:
                self.crf_timestep_history.pop(0)

    # caching.py:561-597
    def predict_crf_freqca(self, t_val: float) -> Optional[torch.Tensor]:
        if not self.use_freqca or self.crf_low_cache is None or len(self.crf_high_history) < 2:
            return None
        from .fourier import predict_hermite

        crf_high_pred = predict_hermite(self.crf_high_history, self.crf_timestep_history, t_val, self.hermite_order)
        if crf_high_pred is None:
            return None
        return self.crf_low_cache + crf_high_pred

    # caching.py:599-653
    def get_cache_stats(self) -> dict:
        rc_count, hit_count, table = self.stats["recompute_count"], self.stats["cache_hit_count"], False
        if self._bound_model is not None:
            st = self._bound_model._native_cache_stats()
            rc_count, hit_count, table = st.recompute_count, st.cache_hit_count, bool(st.table_allocated)
        total = rc_count + hit_count
        ratio = hit_count / total if total > 0 else 0.0
        # cache_valid.float().mean(): after step 0 every row is valid -> capped at 0.99 (caching.py:625-628)
        cache_ratio = 0.99 if table else 0.0
        stats = {"cache_hit_ratio": ratio, "cache_ratio": cache_ratio, "recompute_count": rc_count,
                 "cache_hit_count": hit_count, "current_step": self.current_step}
        if self.use_freqca:  # caching.py:639-651
            n = len(self.crf_high_history)
            stats.update({"freq_decomp_count": n, "freq_decomp_skipped": max(0, self.current_step - n),
                          "freq_decomp_ratio": n / self.current_step if self.current_step > 0 else 0.0})
        return stats
