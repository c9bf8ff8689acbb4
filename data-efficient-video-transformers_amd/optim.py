"""HIP-backed optimizers behind the ``torch.optim.Optimizer`` interface that Lightning's trainer drives
(``configure_optimizers`` of the reference: frame_transformer.py:123-134, transformer.py:58-61).

Per-parameter launches of the fused update kernels (``dvt_adamw_step`` / ``dvt_sgd_step`` /
``dvt_adagrad_step``); a model wrapped in ``dp.FlatParameters`` should use its one-launch
``adamw_step`` / ``sgd_step`` / ``adagrad_step`` instead.  State names match torch's
(``exp_avg``, ``exp_avg_sq``, ``momentum_buffer``, ``sum``, ``step``) so optimizer state dicts
written by the reference load unchanged.
"""
from __future__ import annotations

import torch

from . import ops


def _grad32(p):
    g = p.grad
    if g is None:
        return None
    sink = getattr(p, "_dvt_sink", None)
    if sink is not None and (sink.fresh or sink.unwritten):
        return None        # a FlatParameters view nobody wrote a gradient into: torch would see grad None and skip it
    if g.dtype != torch.float32 or not g.is_contiguous():
        raise RuntimeError("optimizer expects contiguous fp32 gradients (master weights are fp32)")
    return g


class _Base(torch.optim.Optimizer):
    def _invalidate(self, p):
        sink = getattr(p, "_dvt_sink", None)
        if sink is not None:                      # parameters were updated behind a FlatParameters mirror
            sink.owner.invalidate_compute_copy()


class AdamW(_Base):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for grp in self.param_groups:
            for p in grp["params"]:
                g = _grad32(p)
                if g is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                ops.adamw_step_(p.data, g, st["exp_avg"], st["exp_avg_sq"], lr=grp["lr"], beta1=grp["betas"][0],
                                beta2=grp["betas"][1], eps=grp["eps"], weight_decay=grp["weight_decay"],
                                step=st["step"])
                self._invalidate(p)
        return loss


class SGD(_Base):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for grp in self.param_groups:
            for p in grp["params"]:
                g = _grad32(p)
                if g is None:
                    continue
                st = self.state[p]
                if grp["momentum"] != 0 and "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                ops.sgd_step_(p.data, g, st.get("momentum_buffer"), lr=grp["lr"], momentum=grp["momentum"],
                              weight_decay=grp["weight_decay"])
                self._invalidate(p)
        return loss


class Adagrad(_Base):
    def __init__(self, params, lr=1e-2, lr_decay=0.0, weight_decay=0.0, eps=1e-10):
        super().__init__(params, dict(lr=lr, lr_decay=lr_decay, weight_decay=weight_decay, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for grp in self.param_groups:
            for p in grp["params"]:
                g = _grad32(p)
                if g is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["sum"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                ops.adagrad_step_(p.data, g, st["sum"], lr=grp["lr"], lr_decay=grp["lr_decay"], eps=grp["eps"],
                                  weight_decay=grp["weight_decay"], step=st["step"])
                self._invalidate(p)
        return loss
