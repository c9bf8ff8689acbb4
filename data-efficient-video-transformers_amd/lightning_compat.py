"""Minimal stand-in for ``pytorch_lightning.LightningModule`` used when Lightning is not
installed (it is not in the build image).  Only what the reference's modules touch on
the hot path: ``save_hyperparameters`` / ``hparams`` (frame_transformer.py:86-88,
transformer.py:32-34), ``log`` (frame_transformer.py:253-281), ``load_from_checkpoint``
(main.py:89).  With Lightning present the real class is used instead."""
from __future__ import annotations

import inspect
from types import SimpleNamespace

import torch
from torch import nn

try:  # pragma: no cover - not available in the build image
    import pytorch_lightning as _pl
    LightningModule = _pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError in this image
    HAVE_LIGHTNING = False

    class _HParams(SimpleNamespace):
        def __getitem__(self, k):
            return getattr(self, k)

        def __contains__(self, k):
            return hasattr(self, k)

        def get(self, k, default=None):
            return getattr(self, k, default)

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.hparams = _HParams()
            self.logged = {}

        def save_hyperparameters(self, *args, **kwargs):
            """Collects the constructor's keyword arguments of the calling frame, as
            Lightning does for ``Model(**config)`` (main.py:38,44)."""
            frame = inspect.currentframe().f_back
            local = frame.f_locals
            hp = {}
            if "kwargs" in local and isinstance(local["kwargs"], dict):
                hp.update(local["kwargs"])
            for k, v in local.items():
                if k not in ("self", "kwargs", "__class__") and not k.startswith("_"):
                    hp.setdefault(k, v)
            self.hparams = _HParams(**hp)

        def log(self, name, value, **kwargs):
            self.logged[name] = value

        @classmethod
        def load_from_checkpoint(cls, path, **kwargs):
            ckpt = torch.load(path, map_location="cpu")
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"])
            return model
