"""``fdiff.models.score_models`` mirror: ``ScoreModule`` and ``LSTMScoreModule``.

Drop-in for the *sampling* surface of the reference classes
(src/fdiff/models/score_models.py:24-289, 443-511): same constructor arguments,
attributes, ``forward(batch, recompute_tokens, step, return_crf)`` contract,
``enable_caching`` / ``disable_caching`` and -- because the parameter containers are the
same torch modules under the same names -- the same ``state_dict`` keys and default
initialisation, so reference checkpoints load and equal seeds give equal weights.

The torch modules only *hold* parameters.  ``forward`` hands raw device pointers to
libffd (include/ffd.h), whose HIP kernels evaluate the network; nothing here calls a
torch operator on activations, and a tensor that is not on a gfx950 device raises.
Training hooks (training_step, configure_optimizers, losses) are out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Union

import torch
import torch.nn as nn

from .. import _native as N
from ..schedulers.sde import SDE, VEScheduler, VPScheduler
from ..utils.caching import E2CRFCache
from ..utils.dataclasses import DiffusableBatch
from .transformer import GaussianFourierProjection, PositionalEncoding


class _NativeContext:
    """Owns one ffd_ctx (one device) and mirrors the module's parameters into it."""

    def __init__(self, desc: N.ModelDesc, device: torch.device):
        self.lib = N.lib()
        self.handle = C.c_void_p()
        self.device = device
        idx = device.index if device.index is not None else torch.cuda.current_device()
        rc = self.lib.ffd_create(C.byref(self.handle), C.byref(desc), idx)
        if rc != 0:
            try:
                N.check(rc, self.handle, "ffd_create")
            finally:
                if self.handle:
                    self.lib.ffd_destroy(self.handle)
                    self.handle = C.c_void_p()
        self.weights_key = None

    def upload(self, named_tensors) -> None:
        for name, t in named_tensors:
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(torch.float32).contiguous()
            rc = self.lib.ffd_load_weight(self.handle, name.encode(), t.data_ptr(), t.numel())
            N.check(rc, self.handle, f"ffd_load_weight({name})")
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        N.check(self.lib.ffd_finalize_weights(self.handle), self.handle, "ffd_finalize_weights")

    def __del__(self):
        try:
            if self.handle:
                self.lib.ffd_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


class ScoreModule(nn.Module):
    """Transformer score model (score_models.py:24-289)."""

    _kind = N.FFD_MODEL_TRANSFORMER

    def __init__(self, n_channels: int, max_len: int, noise_scheduler: SDE, fourier_noise_scaling: bool = True,
                 d_model: int = 60, num_layers: int = 3, n_head: int = 12, num_training_steps: int = 1000,
                 lr_max: float = 1e-3, likelihood_weighting: bool = False) -> None:
        super().__init__()
        if not isinstance(noise_scheduler, SDE):
            # score_models.py:348-351, 359-361
            raise NotImplementedError(f"Scheduler {noise_scheduler} not implemented yet")
        self.max_len = max_len
        self.n_channels = n_channels
        self.noise_scheduler = noise_scheduler
        self.num_warmup_steps = num_training_steps // 10
        self.num_training_steps = num_training_steps
        self.lr_max = lr_max
        self.d_model = d_model
        self.scale_noise = fourier_noise_scaling
        self.likelihood_weighting = likelihood_weighting
        self.n_head = n_head
        self.num_layers = num_layers

        # parameter containers, constructed in the reference's order (score_models.py:55-66)
        self.pos_encoder = PositionalEncoding(d_model=d_model, max_len=self.max_len)
        self.time_encoder = GaussianFourierProjection(d_model=self.d_model)
        self.embedder = nn.Linear(in_features=n_channels, out_features=d_model)
        self.unembedder = nn.Linear(in_features=d_model, out_features=n_channels)
        self._build_backbone()

        self.cache: Optional[E2CRFCache] = None
        self.use_cache: bool = False
        self.cached_backbone = None  # truthy once enable_caching ran (score_models.py:232)
        self._first_cache: Optional[E2CRFCache] = None
        self._native: Optional[_NativeContext] = None

    def _build_backbone(self) -> None:
        layer = nn.TransformerEncoderLayer(d_model=self.d_model, nhead=self.n_head, batch_first=True)
        self.backbone = nn.TransformerEncoder(encoder_layer=layer, num_layers=self.num_layers)
        self.dim_feedforward = layer.linear1.out_features

    # ------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict: bool = True,
                             weights_only=None, **kwargs):
        """LightningModule.load_from_checkpoint as the reference's runners call it
        (cmd/sample.py:68-75, cmd/benchmark_cache.py:141-144): rebuild the module from the
        checkpoint's ``hyper_parameters`` (saved by ``save_hyperparameters``, score_models.py:71)
        and load its ``state_dict``.  The file is always read with ``torch.load(weights_only=True)``
        (the scheduler classes are allow-listed), whatever ``weights_only`` the caller passes;
        ``cached_backbone.*`` entries (copies made by enable_caching) are ignored."""
        import inspect

        from ..schedulers import sde as _sde

        safe = [_sde.VPScheduler, _sde.VEScheduler]
        with torch.serialization.safe_globals(safe):
            ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(kwargs)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        model = cls(**{k: v for k, v in hp.items() if k in accepted})
        sd = {k: v for k, v in ckpt["state_dict"].items() if not k.startswith("cached_backbone.")}
        model.load_state_dict(sd, strict=strict)
        if map_location is not None:
            model = model.to(map_location)
        return model

    @property
    def device(self) -> torch.device:
        try:
            return next(self.parameters()).device
        except StopIteration:  # pragma: no cover
            return torch.device("cpu")

    def _weight_items(self):
        skip = ()
        for name, p in self.state_dict().items():
            if name.startswith(skip):
                continue
            yield name, p

    def _desc(self) -> N.ModelDesc:
        sch = self.noise_scheduler
        a, b = sch._sde_ab()
        return N.ModelDesc(self._kind, self.n_channels, self.max_len, self.d_model,
                           self.n_head if self._kind == N.FFD_MODEL_TRANSFORMER else 1, self.num_layers,
                           getattr(self, "dim_feedforward", 0), sch._sde_kind, float(a), float(b),
                           int(bool(sch.noise_scaling)), float(sch.eps))

    def _ctx(self) -> _NativeContext:
        dev = self.device
        if dev.type != "cuda":
            raise N.FFDError(
                f"model parameters live on {dev}: fastfourierdiffusion_amd evaluates the score network only on an "
                "MI355X (gfx950) device; call .cuda() (there is no CPU fallback).")
        params = list(self._weight_items())
        key = (str(dev), tuple((n, p.data_ptr(), p._version) for n, p in params),
               self.noise_scheduler._sde_ab(), bool(self.noise_scheduler.noise_scaling))
        if self._native is None or self._native.device != dev or self._native.weights_key is None or \
                self._native.weights_key[2:] != key[2:]:
            self._native = _NativeContext(self._desc(), dev)
            if self.use_cache:
                self._native_cache_enable()
        if self._native.weights_key != key:
            self._native.upload(params)
            self._native.weights_key = key
        return self._native

    # ------------------------------------------------------------------
    def forward(self, batch: DiffusableBatch, recompute_tokens: Optional[set] = None, step: int = 0,
                return_crf: bool = False) -> Union[torch.Tensor, tuple]:
        """score_models.py:79-119."""
        X = batch.X
        assert X.size()[1:] == (self.max_len, self.n_channels), \
            f"X has wrong shape, should be {(X.size(0), self.max_len, self.n_channels)}, but is {X.size()}"
        timesteps = batch.timesteps
        assert timesteps is not None and timesteps.size(0) == len(batch)
        X = N.require_gpu_tensor(X, "batch.X")
        ctx = self._ctx()
        # per-sample diffusion times, as time_encoder(X, timesteps) evaluates them (score_models.py:102;
        # tests/test_score_models.py:70 passes mixed timesteps); they stay on the device -- no host sync
        ts = timesteps.to(device=X.device, dtype=torch.float32).contiguous()
        B = X.shape[0]
        score = torch.empty_like(X)
        stream = N.current_stream_ptr(X.device)
        cached = self.use_cache and recompute_tokens is not None and self.cached_backbone is not None
        if not cached:
            N.check(ctx.lib.ffd_score_forward_ts(ctx.handle, X.data_ptr(), ts.data_ptr(), score.data_ptr(), None, B, -1,
                                                 stream), ctx.handle, "ffd_score_forward_ts")
            return (score, None) if return_crf else score
        n = len(recompute_tokens)
        if set(recompute_tokens) != set(range(n)):
            raise NotImplementedError(
                "recompute_tokens must be a prefix {0..n-1}: the only sets E2CRFCache.determine_recompute_set "
                "produces (caching.py:131-181)")
        crf = torch.empty((self.num_layers, self.max_len, self.d_model), device=X.device, dtype=torch.float32) \
            if return_crf else None
        N.check(ctx.lib.ffd_score_forward_ts(ctx.handle, X.data_ptr(), ts.data_ptr(), score.data_ptr(),
                                             crf.data_ptr() if crf is not None else None, B, n, stream),
                ctx.handle, "ffd_score_forward_ts")
        return (score, crf) if return_crf else score

    # ------------------------------------------------------------------
    def enable_caching(self, cache: Optional[E2CRFCache] = None, **cache_kwargs) -> None:
        """score_models.py:202-283.  A new E2CRFCache is created on every call, but the
        layers stay bound to the cache of the *first* call (Q5)."""
        if self._kind != N.FFD_MODEL_TRANSFORMER:
            # the reference dereferences backbone.layers, which ModuleList lacks (Q9)
            raise AttributeError("'ModuleList' object has no attribute 'layers'")
        if cache is None:
            cache = E2CRFCache(max_len=self.max_len, num_layers=self.num_layers, device=self.device, **cache_kwargs)
        self.cache = cache
        self.use_cache = True
        if self.cached_backbone is None:
            self.cached_backbone = True
            self._first_cache = cache
            cache._bound_model = self
            if self._native is not None:
                self._native_cache_enable()

    def disable_caching(self) -> None:
        """score_models.py:285-288."""
        self.use_cache = False
        self.cache = None

    def _native_cache_enable(self) -> None:
        fc = self._first_cache
        cfg = N.CacheCfg(int(fc.K), int(fc.R)) if fc is not None else N.CacheCfg(5, 10)
        N.check(self._native.lib.ffd_cache_enable(self._native.handle, C.byref(cfg)), self._native.handle,
                "ffd_cache_enable")

    def _native_cache_configure(self, cache: E2CRFCache) -> None:
        """Gate parameters of the sampler's *current* cache (the tables stay bound to the first one, Q5)."""
        if self._native is not None:
            cfg = N.CacheCfg(int(cache.K), int(cache.R))
            N.check(self._native.lib.ffd_cache_configure(self._native.handle, C.byref(cfg)), self._native.handle,
                    "ffd_cache_configure")

    def _native_cache_reset(self) -> None:
        if self._native is not None:
            N.check(self._native.lib.ffd_cache_reset(self._native.handle), self._native.handle, "ffd_cache_reset")

    def _native_cache_stats(self) -> N.CacheStats:
        st = N.CacheStats()
        if self._native is not None:
            N.check(self._native.lib.ffd_cache_stats_get(self._native.handle, C.byref(st)), self._native.handle,
                    "ffd_cache_stats_get")
        return st

    def cache_tables(self):
        """(K, V) tables (NL, H, L, hd) copied into fresh device tensors (caching.py:88-91)."""
        ctx = self._ctx()
        shape = (self.num_layers, self.n_head, self.max_len, self.d_model // self.n_head)
        k = torch.empty(shape, device=self.device, dtype=torch.float32)
        v = torch.empty(shape, device=self.device, dtype=torch.float32)
        N.check(ctx.lib.ffd_cache_tables_read(ctx.handle, k.data_ptr(), v.data_ptr(),
                                              N.current_stream_ptr(self.device)), ctx.handle, "ffd_cache_tables_read")
        return k, v


class LSTMScoreModule(ScoreModule):
    """Residual-LSTM score model (score_models.py:443-511): no positional encoding,
    ``x <- x + LSTM_l(x)`` for each of ``num_layers`` nn.LSTM(d, d) layers."""

    _kind = N.FFD_MODEL_LSTM

    def __init__(self, n_channels: int, max_len: int, noise_scheduler: SDE, fourier_noise_scaling: bool = True,
                 d_model: int = 72, num_layers: int = 3, num_training_steps: int = 1000, lr_max: float = 1e-3,
                 likelihood_weighting: bool = False) -> None:
        super().__init__(n_channels=n_channels, max_len=max_len, noise_scheduler=noise_scheduler,
                         fourier_noise_scaling=fourier_noise_scaling, d_model=d_model, num_layers=num_layers, n_head=1,
                         num_training_steps=num_training_steps, lr_max=lr_max,
                         likelihood_weighting=likelihood_weighting)

    def _build_backbone(self) -> None:
        # the reference first builds (and then discards) a 1-head TransformerEncoder in
        # ScoreModule.__init__ (score_models.py:458-468), consuming RNG draws; replay that
        # so equal seeds give equal LSTM weights.
        layer = nn.TransformerEncoderLayer(d_model=self.d_model, nhead=1, batch_first=True)
        nn.TransformerEncoder(encoder_layer=layer, num_layers=self.num_layers)
        self.backbone = nn.ModuleList([
            nn.LSTM(input_size=self.d_model, hidden_size=self.d_model, batch_first=True, bidirectional=False)
            for _ in range(self.num_layers)
        ])
        self.pos_encoder = None
        self.dim_feedforward = 0

    def forward(self, batch: DiffusableBatch) -> torch.Tensor:  # type: ignore[override]
        """score_models.py:486-511 (no recompute_tokens / caching, Q9)."""
        return super().forward(batch)


class MLPScoreModule(ScoreModule):
    """MLP score model (score_models.py:363-440): the flattened series (L*C) is embedded to d_model,
    ``x <- x + MLP_l(x)`` for ``num_layers`` blocks d -> d_mlp -> d (ReLU), then unembedded to L*C.
    The reference's block is ``torchvision.ops.MLP`` (= Sequential(Linear, ReLU, Dropout, Linear, Dropout));
    the same container layout is built here so the state_dict keys (``backbone.{i}.0.*``, ``backbone.{i}.3.*``)
    match."""

    _kind = N.FFD_MODEL_MLP

    def __init__(self, n_channels: int, max_len: int, noise_scheduler: SDE, fourier_noise_scaling: bool = True,
                 d_model: int = 72, d_mlp: int = 512, num_layers: int = 3, num_training_steps: int = 1000,
                 lr_max: float = 1e-3, likelihood_weighting: bool = False) -> None:
        self.d_mlp = d_mlp  # plain attribute: needed by _build_backbone, which ScoreModule.__init__ calls
        super().__init__(n_channels=n_channels, max_len=max_len, noise_scheduler=noise_scheduler,
                         fourier_noise_scaling=fourier_noise_scaling, d_model=d_model, num_layers=num_layers, n_head=1,
                         num_training_steps=num_training_steps, lr_max=lr_max,
                         likelihood_weighting=likelihood_weighting)

    def _build_backbone(self) -> None:
        # replay the reference's construction order (a discarded 1-head TransformerEncoder from
        # ScoreModule.__init__, then embedder / unembedder / blocks, score_models.py:377-403)
        layer = nn.TransformerEncoderLayer(d_model=self.d_model, nhead=1, batch_first=True)
        nn.TransformerEncoder(encoder_layer=layer, num_layers=self.num_layers)
        io = self.max_len * self.n_channels
        self.embedder = nn.Linear(in_features=io, out_features=self.d_model)
        self.unembedder = nn.Linear(in_features=self.d_model, out_features=io)
        self.backbone = nn.ModuleList([
            nn.Sequential(nn.Linear(self.d_model, self.d_mlp), nn.ReLU(), nn.Dropout(0.1),
                          nn.Linear(self.d_mlp, self.d_model), nn.Dropout(0.1))
            for _ in range(self.num_layers)
        ])
        self.pos_encoder = None
        self.dim_feedforward = self.d_mlp

    def forward(self, batch: DiffusableBatch) -> torch.Tensor:  # type: ignore[override]
        """score_models.py:406-440 (no recompute_tokens / caching, Q9)."""
        return super().forward(batch)
