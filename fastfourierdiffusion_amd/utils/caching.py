"""``fdiff.utils.caching`` mirror: ``E2CRFCache`` (reference caching.py:19-653).

Host-side state machine only: the gate (a pure function of the step index), the
counters and the CRF slot.  The K/V tables (NL, H, L, hd) themselves live in the
native context's HBM (csrc/ffd_api.hip) and are read by the attention kernel; the
cache object that a model's layers were *first* bound to is the one whose ``reset``
and statistics reach them (reference quirk Q5, score_models.py:232).
"""
from __future__ import annotations

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
        if use_freqca:
            raise NotImplementedError("use_freqca=True (FreqCa CRF decomposition) is a 'next' row of the scope table")
        self.crf_cache: Optional[torch.Tensor] = None
        self.stats = {"recompute_count": 0, "cache_hit_count": 0}
        self.current_step = 0
        self._bound_model = None  # the score model whose native tables this object controls

    # caching.py:115-129
    def reset(self) -> None:
        self.crf_cache = None
        self.stats = {"recompute_count": 0, "cache_hit_count": 0}
        self.current_step = 0
        if self._bound_model is not None:
            self._bound_model._native_cache_reset()

    # caching.py:131-181
    def determine_recompute_set(self, x_tilde, event_intensity: float, step: int) -> set:
        n = N.lib().ffd_host_gate(int(step), int(self.max_len), int(self.K), int(self.R))
        return set(range(n))

    # caching.py:459-484 (FreqCa branch not built)
    def update_crf(self, crf: torch.Tensor, timestep: Optional[float] = None) -> None:
        if self.current_step % self.R == 0:
            self.crf_cache = crf.detach()

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
        return {"cache_hit_ratio": ratio, "cache_ratio": cache_ratio, "recompute_count": rc_count,
                "cache_hit_count": hit_count, "current_step": self.current_step}
