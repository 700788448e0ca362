"""``fdiff.utils.extraction`` mirror (reference src/fdiff/utils/extraction.py:12-121): the checkpoint / config
plumbing ``cmd/sample.py:52-75`` runs before it can sample -- pick the best checkpoint of a run directory, map the
training config's ``score_model._target_`` to a model class, flatten a config for logging.  Pure host code on
plain ``dict`` configs (``omegaconf`` is optional: a ``DictConfig`` is converted when the package is importable).

``instantiate`` is the small part of ``hydra.utils.instantiate`` the sampling runners rely on (``_target_`` dotted
paths, ``_partial_: true``, nested configs), so ``cmd/conf/{sampler,score_model,noise_scheduler}/*.yaml`` resolve
against this package after ``install_as_fdiff()`` also where Hydra itself is not installed.
"""
from __future__ import annotations

import functools
import importlib
import re
from pathlib import Path
from typing import Any, Callable, Union

_CKPT_RE = re.compile(r"(.+?)epoch=(\d+)-val_loss=(\d+\.\d+).ckpt")
_MODEL_TARGETS = {
    "fdiff.models.score_models.ScoreModule": "ScoreModule",
    "fdiff.models.score_models.MLPScoreModule": "MLPScoreModule",
    "fdiff.models.score_models.LSTMScoreModule": "LSTMScoreModule",
}


def _as_dict(cfg: Any) -> dict:
    if isinstance(cfg, dict):
        return cfg
    try:  # a DictConfig, where omegaconf exists
        from omegaconf import DictConfig, OmegaConf

        if isinstance(cfg, DictConfig):
            out = OmegaConf.to_container(cfg, resolve=True)
            assert isinstance(out, dict)
            return out
    except ImportError:
        pass
    raise TypeError(f"config must be a dict (or DictConfig), got {type(cfg)}")


def get_training_params(datamodule: Any, trainer: Any) -> dict:
    """extraction.py:12-17: the datamodule's dataset parameters with num_training_steps scaled to the run."""
    params = datamodule.dataset_parameters
    params["num_training_steps"] *= trainer.max_epochs
    params["num_training_steps"] /= trainer.accumulate_grad_batches
    assert isinstance(params, dict)
    return params


def flatten_config(cfg: Any) -> dict:
    """extraction.py:20-54: nested config -> one flat dict; a sub-config with a ``_target_`` contributes
    ``key: target`` plus its own flattened entries, lists of sub-configs become lists of targets, the
    ``_target_`` / ``_partial_`` markers themselves are dropped."""
    flat: dict = {}
    for key, val in _as_dict(cfg).items():
        if isinstance(val, dict):
            if "_target_" in val:
                flat[key] = val["_target_"]
            flat.update(flatten_config(val))
        elif isinstance(val, list):
            targets = []
            for item in val:
                if isinstance(item, dict):
                    if "_target_" in item:
                        targets.append(item["_target_"])
                    flat.update(flatten_config(item))
            flat[key] = targets
        elif key not in ("_target_", "_partial_"):
            flat[key] = val
    return flat


def get_model_type(cfg: Any):
    """extraction.py:57-76: the model class named by ``cfg["score_model"]["_target_"]``."""
    target = cfg["score_model"]["_target_"]
    if target not in _MODEL_TARGETS:
        raise NotImplementedError(f"Model class {target} not implemented yet.")
    from ..models import score_models

    return getattr(score_models, _MODEL_TARGETS[target])


def get_best_checkpoint(checkpoint_path: Union[str, Path]) -> Path:
    """extraction.py:79-98: among ``*epoch=E-val_loss=X.ckpt`` files of a directory, the one with the lowest
    validation loss (files that do not match the pattern are ignored; none matching raises, as the reference's
    unbound local does)."""
    best, best_loss = None, float("inf")
    for ckpt in Path(checkpoint_path).glob("*.ckpt"):
        m = _CKPT_RE.match(str(ckpt))
        if m is not None and float(m.group(3)) < best_loss:
            best_loss, best = float(m.group(3)), ckpt
    if best is None:
        raise UnboundLocalError(f"no '*epoch=<n>-val_loss=<x>.ckpt' file under {checkpoint_path}")
    return best


def dict_to_str(d: Any) -> str:
    """extraction.py:101-121: one aligned ``key : value`` line per entry; long lists show three elements."""
    if not isinstance(d, dict):
        d = flatten_config(d)
    width = max(len(k) for k in d) + 5
    lines = []
    for k, v in d.items():
        if isinstance(v, list) and len(v) > 3:
            v = v[:3] + ["..."]
        lines.append(f"\t {k: <{width}} : \t  {v} \t \n")
    return "".join(lines)


def _locate(path: str) -> Callable:
    mod, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def instantiate(cfg: Any, **overrides: Any) -> Any:
    """``hydra.utils.instantiate`` for the shapes the sampling configs use: ``_target_`` = dotted path of a
    callable, every other key a keyword argument (nested ``_target_`` dicts are instantiated first),
    ``_partial_: true`` returns ``functools.partial`` instead of calling."""
    cfg = dict(_as_dict(cfg))
    target = cfg.pop("_target_")
    partial = bool(cfg.pop("_partial_", False))
    kwargs = {k: (instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in cfg.items()}
    kwargs.update(overrides)
    fn = _locate(target)
    return functools.partial(fn, **kwargs) if partial else fn(**kwargs)
