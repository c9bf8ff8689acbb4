"""MI355X-native video-clip forward/backward hot path of
ed-fish/data-efficient-video-transformers.

Layout
  csrc/          hand-written HIP kernels (gfx950) + the C ABI  -> libdvt_hip.so
  _lib.py        ctypes binding of include/dvt_hip.h
  ops.py         raw operator wrappers (no autograd)
  functional.py  torch.autograd.Function layer (fused residual blocks)
  models/        mirror of the reference's src/models surface (vit.py, ...)
  dp.py          data-parallel gradient all-reduce over RCCL

The directory name is the one the task prescribes and is not a Python
identifier; import the package through the alias module ``dvt_amd`` at the
repository root (``import dvt_amd``).
"""
from . import _lib, ops, functional  # noqa: F401
from .build import build as build_extension  # noqa: F401

__all__ = ["_lib", "ops", "functional", "models", "build_extension"]
