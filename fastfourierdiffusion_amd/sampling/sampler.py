"""``fdiff.sampling.sampler`` mirror: ``DiffusionSampler``
(reference src/fdiff/sampling/sampler.py:14-228).

Same constructor, attributes and ``sample`` / ``reverse_diffusion_step`` /
``sample_prior`` contracts.  ``sample`` keeps the reference's batching rules (remainder
samples dropped, Q10), cache lifecycle (reset only for batch 0, global step keeps
counting, Q3) and returns a CPU tensor, but runs each batch's whole reverse-diffusion
loop through ``ffd_sample_batch``: every kernel of every step is enqueued on the
current HIP stream with no host synchronisation (the reference syncs several times per
step: ``.item()``, ``min == max`` assert, H2D ``diag_embed`` copies).

Noise: ``rng="torch"`` (default) draws z exactly where the reference does --
``torch.randn`` on the CPU generator for the prior, ``torch.randn_like`` on the device
generator for every step -- so a seeded run matches the reference run on the same
device.  ``rng="philox"`` generates the draws inside the step kernel (Philox4x32-10
keyed by (seed, step, global element index)): no z traffic, and the NOISE is invariant
to how samples are sharded over GPUs.

Two batch-wide statistics of the reference make the SAMPLES depend on the batching all the
same, sharded or not (they are properties of the reference's algorithm, reproduced here per
batch / per shard): the E2-CRF tables come from the batch's element 0 (caching.py:326-328), and
FreSca's default ``energy`` cutoff is derived from ``|rfft(score)|.mean(dim=(0, 2))`` over the
local batch (fresca.py:150-158).  Without the cache and with FreSca off or on its ``spatial``
cutoff a sharded philox run reproduces the unsharded one: the same noise bit for bit, the samples to fp32 rounding
(a few 1e-7 relative per score evaluation; the contract the tests hold is 1e-5 of the max-norm over a trajectory) --
not bit for bit, because the kernels a batch size selects differ in summation order (the F-split small-batch FFN pair,
the key-split attention, the 32-row-per-wave large-batch FFN, the batch-tiled LSTM recurrence; csrc/ffd_small.hip,
ffd_ffn_rows.hip), and a shard may fall into another regime than the whole batch (tests/test_gpu_parity.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Literal, Optional

import torch

from .. import _native as N
from ..models.score_models import ScoreModule
from ..schedulers.sde import SDE
from ..utils.dataclasses import DiffusableBatch


class DiffusionSampler:
    def __init__(self, score_model: ScoreModule, sample_batch_size: int, use_cache: bool = False,
                 cache_kwargs: Optional[dict] = None, use_fresca: bool = False, fresca_low_scale: float = 1.0,
                 fresca_high_scale: float = 1.5, fresca_cutoff_ratio: float = 0.5,
                 fresca_cutoff_strategy: Literal["spatial", "energy"] = "energy",
                 rng: Literal["torch", "philox"] = "torch", seed: int = 42, sample_offset: int = 0,
                 z_chunk_steps: int = 50) -> None:
        self.score_model = score_model
        self.noise_scheduler = score_model.noise_scheduler
        self.sample_batch_size = sample_batch_size
        self.n_channels = score_model.n_channels
        self.max_len = score_model.max_len

        self.use_cache = use_cache
        if use_cache:
            cache_kwargs = cache_kwargs or {}
            self.score_model.enable_caching(**cache_kwargs)  # sampler.py:37-39

        self.use_fresca = use_fresca
        self.fresca_low_scale = fresca_low_scale
        self.fresca_high_scale = fresca_high_scale
        self.fresca_cutoff_ratio = fresca_cutoff_ratio
        self.fresca_cutoff_strategy = fresca_cutoff_strategy
        if fresca_cutoff_strategy not in ("spatial", "energy"):
            self.fresca_cutoff_strategy = "spatial"  # sampler.py:83-85 maps anything but "energy" to "spatial"
        # extensions (not in the reference signature)
        assert rng in ("torch", "philox")
        self.rng = rng
        self.seed = int(seed)
        self.sample_offset = int(sample_offset)
        self.z_chunk_steps = int(z_chunk_steps)
        self._injected = None

    def inject_noise(self, draws) -> None:
        """Parity hook: take every N(0,1) draw (one (B,L,C) array for each batch's prior,
        then one per step) from the iterable ``draws`` instead of torch's generators --
        the product-side twin of the noise patching in oracle/gen_golden.py."""
        self._injected = iter(draws) if draws is not None else None

    def _next_injected(self, shape, device) -> torch.Tensor:
        z = next(self._injected)
        z = torch.as_tensor(z, dtype=torch.float32)
        assert tuple(z.shape) == tuple(shape), (tuple(z.shape), tuple(shape))
        return z.to(device)

    # ------------------------------------------------------------------
    def reverse_diffusion_step(self, batch: DiffusableBatch, step: int = 0,
                               recompute_tokens: Optional[set] = None) -> torch.Tensor:
        """sampler.py:48-103 (single-step API; ``sample`` uses the fused loop)."""
        X = batch.X
        timesteps = batch.timesteps
        assert timesteps is not None and timesteps.size(0) == len(batch)
        t_lo, t_hi = float(torch.min(timesteps)), float(torch.max(timesteps))
        assert t_lo == t_hi  # sampler.py:59-60
        if self.use_cache and recompute_tokens is not None:
            score, crf = self.score_model(batch, recompute_tokens=recompute_tokens, step=step, return_crf=True)
            if self.score_model.cache is not None:
                self.score_model.cache.update_crf(crf, timestep=t_lo)
                self.score_model.cache.current_step = step
        else:
            score = self.score_model(batch)
        if self.use_fresca:  # sampler.py:79-93
            from ..utils.fresca import apply_fresca_to_score

            score = apply_fresca_to_score(score, low_scale=self.fresca_low_scale, high_scale=self.fresca_high_scale,
                                          cutoff_ratio=self.fresca_cutoff_ratio,
                                          cutoff_strategy="energy" if self.fresca_cutoff_strategy == "energy" else "spatial",
                                          timestep=t_lo, num_steps=getattr(self, "_num_diffusion_steps", None))
        output = self.noise_scheduler.step(model_output=score, timestep=timesteps[0].item(), sample=X)
        X_prev = output.prev_sample
        assert isinstance(X_prev, torch.Tensor)
        return X_prev

    # ------------------------------------------------------------------
    def sample(self, num_samples: int, num_diffusion_steps: Optional[int] = None) -> torch.Tensor:
        """sampler.py:105-215."""
        if num_diffusion_steps is not None:
            self._num_diffusion_steps = num_diffusion_steps
        self.score_model.eval()
        num_diffusion_steps = (self.score_model.num_training_steps if num_diffusion_steps is None
                               else num_diffusion_steps)
        self.noise_scheduler.set_timesteps(num_diffusion_steps)
        sch = self.noise_scheduler
        ts = sch.timesteps.to(torch.float32).contiguous()
        ts_c = (C.c_float * num_diffusion_steps)(*ts.tolist())
        step_size = float(sch.step_size)

        model = self.score_model
        ctx = model._ctx()
        if self.use_fresca:
            cfg = N.FrescaCfg(float(self.fresca_low_scale), float(self.fresca_high_scale),
                              float(self.fresca_cutoff_ratio),
                              1 if self.fresca_cutoff_strategy == "energy" else 0,
                              int(getattr(self, "_num_diffusion_steps", 0) or 0))
            N.check(ctx.lib.ffd_fresca_enable(ctx.handle, C.byref(cfg)), ctx.handle, "ffd_fresca_enable")
        else:
            N.check(ctx.lib.ffd_fresca_disable(ctx.handle), ctx.handle, "ffd_fresca_disable")
        device = model.device
        stream = N.current_stream_ptr(device)
        all_samples = []
        num_batches = max(1, num_samples // self.sample_batch_size)  # remainder dropped (Q10)
        global_step = 0
        sample_cursor = self.sample_offset
        with torch.no_grad():
            for batch_idx in range(num_batches):
                batch_size = min(num_samples - batch_idx * self.sample_batch_size, self.sample_batch_size)
                X = self.sample_prior(batch_size, _sample_offset=sample_cursor)
                if self.use_cache and model.cache is not None and batch_idx == 0:
                    model.cache.reset()  # sampler.py:151-153
                    global_step = 0
                use_cache = int(self.use_cache and model.cache is not None)
                if use_cache:  # the gate reads K, R of the current cache object (sampler.py:179-200)
                    model._native_cache_configure(model.cache)
                cap = self._crf_capture_begin(ctx, device) if use_cache else None
                done = 0
                while done < num_diffusion_steps:
                    n = num_diffusion_steps - done
                    z_ptr = None
                    if self.rng == "torch" or self._injected is not None:
                        n = min(n, max(1, self.z_chunk_steps))
                        z = torch.empty((n,) + tuple(X.shape), device=device, dtype=torch.float32)
                        for i in range(n):  # one randn_like per step, as the reference consumes its generator
                            if self._injected is not None:
                                z[i].copy_(self._next_injected(X.shape, device))
                            else:
                                z[i].normal_()
                        z_ptr = z.data_ptr()
                    rc = ctx.lib.ffd_sample_batch(ctx.handle, X.data_ptr(), batch_size, ts_c, num_diffusion_steps,
                                                  step_size, done, n, self.seed, sample_cursor, z_ptr, use_cache,
                                                  (global_step + done) if use_cache else 0, stream)
                    N.check(rc, ctx.handle, "ffd_sample_batch")
                    if cap is not None:
                        self._crf_capture_collect(cap, global_step + done, n, ts, done)
                    done += n
                if use_cache:
                    N.check(ctx.lib.ffd_cache_crf_capture(ctx.handle, None), ctx.handle, "ffd_cache_crf_capture")
                    global_step += num_diffusion_steps
                    model.cache.current_step = num_diffusion_steps - 1  # sampler.py:73-74 leaves step_idx
                all_samples.append(X.cpu())
                # (the copy has drained the stream) a kernel-side time-out of this batch's launches is an error here
                N.check(ctx.lib.ffd_async_status(ctx.handle), ctx.handle, "sampling loop")
                sample_cursor += batch_size
        return torch.cat(all_samples, dim=0)

    # ------------------------------------------------------------------
    # cache.update_crf (sampler.py:70-73) for the fused loop: the native loop writes the CRF of the steps the
    # reference would have stored (E2CRFCache.update_crf, caching.py:474-522) into these buffers.
    def _crf_capture_begin(self, ctx, device):
        cache = self.score_model.cache
        m = self.score_model
        shape = (m.num_layers, m.max_len, m.d_model)
        last = torch.empty(shape, device=device, dtype=torch.float32)
        ring = None
        if cache.use_freqca:
            ring = torch.empty((max(1, int(cache.max_history)),) + shape, device=device, dtype=torch.float32)
        cfg = N.CrfCaptureCfg(ring.data_ptr() if ring is not None else None, ring.shape[0] if ring is not None else 0,
                              max(1, int(cache.freq_decomp_interval)) if ring is not None else 0, last.data_ptr(),
                              1 if cache.use_freqca else max(1, int(cache.R)), 0)
        N.check(ctx.lib.ffd_cache_crf_capture(ctx.handle, C.byref(cfg)), ctx.handle, "ffd_cache_crf_capture")
        return {"last": last, "ring": ring, "cfg": cfg}

    def _crf_capture_collect(self, cap, g0: int, n: int, ts: torch.Tensor, step0: int) -> None:
        """Fold what ffd_sample_batch captured over global steps [g0, g0+n) into the cache object."""
        cache = self.score_model.cache
        cfg = cap["cfg"]
        if any(g % cfg.last_every == 0 for g in range(g0, g0 + n)):
            cache.crf_cache = cap["last"].clone()
        if cap["ring"] is not None:
            from ..utils.fourier import frequency_decompose_dct, frequency_decompose_fft

            fn = frequency_decompose_fft if cache.freq_decomp == "fft" else frequency_decompose_dct
            hits = [g for g in range(g0, g0 + n) if g % cfg.every == 0][-cfg.n_slots:]
            for g in hits:  # only the newest max_history decompositions can survive in the history
                slot = (g // cfg.every) % cfg.n_slots
                low, high = fn(cap["ring"][slot], cache.low_freq_ratio)
                cache._push_decomposition(low, high, float(ts[step0 + (g - g0)]))

    def sample_prior(self, batch_size: int, _sample_offset: int = 0) -> torch.Tensor:
        """sampler.py:217-228."""
        if not isinstance(self.noise_scheduler, SDE):
            raise NotImplementedError("Scheduler not recognized.")
        device = self.score_model.device
        shape = (batch_size, self.max_len, self.n_channels)
        if self._injected is not None:
            if device.type != "cuda":
                raise N.FFDError("sampling needs an MI355X (gfx950) device; there is no CPU fallback")
            sch = self.noise_scheduler
            z = self._next_injected(shape, device)
            X = torch.empty(shape, device=device, dtype=torch.float32)
            desc = sch._desc()
            rc = N.lib().ffd_prior(C.byref(desc), X.data_ptr(), z.data_ptr(), sch._G_on(device).data_ptr(), 0, 0,
                                   batch_size, self.max_len, self.n_channels, N.current_stream_ptr(device))
            N.check(rc, None, "ffd_prior")
        elif self.rng == "torch":
            X = self.noise_scheduler.prior_sampling(shape, device=device)
        else:
            if device.type != "cuda":
                raise N.FFDError("sampling needs an MI355X (gfx950) device; there is no CPU fallback")
            sch = self.noise_scheduler
            X = torch.empty(shape, device=device, dtype=torch.float32)
            desc = sch._desc()
            rc = N.lib().ffd_prior(C.byref(desc), X.data_ptr(), None, sch._G_on(device).data_ptr(), self.seed,
                                   _sample_offset, batch_size, self.max_len, self.n_channels,
                                   N.current_stream_ptr(device))
            N.check(rc, None, "ffd_prior")
        assert isinstance(X, torch.Tensor)
        return X
