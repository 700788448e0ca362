"""TEST INFRASTRUCTURE ONLY -- loader for the *unmodified* reference package.

Used exclusively by ``oracle/gen_golden.py`` (fixture generation, in the build
container, where ``/root/reference`` is mounted) and by the optional
``tests/test_oracle_vs_reference.py`` cross-check.  Nothing here is shipped or
imported by the product package ``fastfourierdiffusion_amd``.

The reference's ``fdiff.models.score_models`` and ``fdiff.sampling.sampler``
fail to import in this image only because three third-party packages are not
installed (``pytorch_lightning``, ``diffusers``, ``torchvision``; see
/root/reference/src/fdiff/models/score_models.py:3-10).  None of them touches
the sampling arithmetic: ``LightningModule`` contributes ``.device`` and no-op
logging hooks, ``get_cosine_schedule_with_warmup`` is a training LR schedule,
``torchvision.ops.MLP`` is only used by ``MLPScoreModule`` (out of scope).
We register three inert stub modules in ``sys.modules`` and then import the
reference from where it lies.  No reference source is copied.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_SRC = os.environ.get("FFD_REFERENCE_SRC", "/root/reference/src")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "fdiff"))


def _install_stubs() -> None:
    import torch
    import torch.nn as nn

    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(nn.Module):
            def save_hyperparameters(self, *a, **k):
                return None

            def log_dict(self, *a, **k):
                return None

            def log(self, *a, **k):
                return None

            @property
            def device(self):
                try:
                    return next(self.parameters()).device
                except StopIteration:
                    return torch.device("cpu")

        class Callback:
            pass

        class LightningDataModule:
            pass

        class Trainer:
            pass

        pl.LightningModule = LightningModule
        pl.Callback = Callback
        pl.LightningDataModule = LightningDataModule
        pl.Trainer = Trainer
        util = types.ModuleType("pytorch_lightning.utilities")
        tps = types.ModuleType("pytorch_lightning.utilities.types")
        tps.OptimizerLRScheduler = object
        util.types = tps
        pl.utilities = util
        sys.modules["pytorch_lightning"] = pl
        sys.modules["pytorch_lightning.utilities"] = util
        sys.modules["pytorch_lightning.utilities.types"] = tps

    if "diffusers" not in sys.modules:
        df = types.ModuleType("diffusers")
        opt = types.ModuleType("diffusers.optimization")

        def get_cosine_schedule_with_warmup(*a, **k):
            raise NotImplementedError("training-only; stubbed for sampling oracle")

        opt.get_cosine_schedule_with_warmup = get_cosine_schedule_with_warmup
        df.optimization = opt
        sys.modules["diffusers"] = df
        sys.modules["diffusers.optimization"] = opt

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        ops = types.ModuleType("torchvision.ops")
        ops.MLP = object
        tv.ops = ops
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = ops


def import_reference():
    """Return a namespace with the reference classes used by the sampling path."""
    if not reference_available():
        raise RuntimeError(f"reference sources not found under {REFERENCE_SRC}")
    os.environ.setdefault("TQDM_DISABLE", "1")
    _install_stubs()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    ns = types.SimpleNamespace()
    from fdiff.schedulers.sde import VEScheduler, VPScheduler, SDE
    from fdiff.utils.fourier import dft, idft
    from fdiff.utils.dataclasses import DiffusableBatch
    from fdiff.utils.caching import E2CRFCache
    from fdiff.models.transformer import GaussianFourierProjection, PositionalEncoding
    from fdiff.models.cached_transformer import CachedTransformerEncoderLayer
    from fdiff.models.score_models import ScoreModule, LSTMScoreModule
    from fdiff.sampling.sampler import DiffusionSampler

    ns.VEScheduler, ns.VPScheduler, ns.SDE = VEScheduler, VPScheduler, SDE
    ns.dft, ns.idft = dft, idft
    ns.DiffusableBatch = DiffusableBatch
    ns.E2CRFCache = E2CRFCache
    ns.GaussianFourierProjection = GaussianFourierProjection
    ns.PositionalEncoding = PositionalEncoding
    ns.CachedTransformerEncoderLayer = CachedTransformerEncoderLayer
    ns.ScoreModule, ns.LSTMScoreModule = ScoreModule, LSTMScoreModule
    ns.DiffusionSampler = DiffusionSampler
    return ns
