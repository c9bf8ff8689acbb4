"""Evaluation-side reductions on the device and the callback that consumes them.

Mirror of ``TransformerEval`` (src/callbacks/callbacks.py:27-66): at the end of a validation epoch the
accumulated ``pl_module.running_logits`` / ``running_labels`` are reduced to the samples-F1 threshold sweep and
the two average-precision scores, logged under the reference's keys, and the accumulators are reset.  The
reference moves everything to the host and calls scikit-learn; here the tensors stay in HBM, and under data
parallelism every rank first all-gathers the other ranks' accumulators (the reference is single-GPU).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops

THRESHOLDS = (0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8)          # callbacks.py:38


def gather_rows(t: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Concatenate every rank's rows (ranks may hold different row counts)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(counts)
    pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def evaluate(probs: torch.Tensor, labels: torch.Tensor, thresholds=THRESHOLDS) -> Dict[str, float]:
    """probs [N, C] (sigmoid outputs), labels [N, C] -> the reference's logged scalars."""
    f1 = ops.f1_samples(probs, labels, thresholds)
    ap_s, ap_w, _ = ops.average_precision(probs, labels)
    host = torch.cat((f1, ap_s, ap_w)).cpu()                      # one device->host copy per epoch
    out = {f"val/online/f1@{str(t)}": float(host[i]) for i, t in enumerate(thresholds)}
    out["sklearn apr"] = float(host[len(thresholds)])
    out["sklearn apr weighted"] = float(host[len(thresholds) + 1])
    return out


class TransformerEval:
    """callbacks.py:27-66 (``on_validation_epoch_end``); same log keys, same accumulator reset."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group

    def on_validation_epoch_end(self, trainer, pl_module) -> Dict[str, float]:
        labels = gather_rows(torch.cat(pl_module.running_labels), self.group)
        probs = gather_rows(torch.cat(pl_module.running_logits), self.group)
        scalars = evaluate(probs, labels)
        for k, v in scalars.items():
            pl_module.log(k, v)
        pl_module.running_labels = []
        pl_module.running_logits = []
        return scalars


class AveragePrecision:
    """Stand-in for ``torchmetrics.AveragePrecision(num_classes=C)`` (frame_transformer.py:114,118,277,335): calling the
    object with ``(preds, target)`` accumulates a batch on the device, ``compute()`` returns the per-class average
    precision (one-vs-rest, a list of C scalars like torchmetrics 0.6) and ``reset()`` clears it.  Average precision is
    rank-based, so logits and probabilities give the same value."""

    def __init__(self, num_classes: int):
        self.num_classes = num_classes
        self.preds: List[torch.Tensor] = []
        self.target: List[torch.Tensor] = []

    def __call__(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        self.update(preds, target)

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        if preds.shape[-1] != self.num_classes:
            raise ValueError(f"expected {self.num_classes} classes")
        self.preds.append(preds.detach().reshape(-1, self.num_classes))
        self.target.append(target.detach().reshape(-1, self.num_classes))

    def compute(self, group: Optional[dist.ProcessGroup] = None) -> List[torch.Tensor]:
        if not self.preds:
            raise RuntimeError("AveragePrecision.compute() before any update")
        p = gather_rows(torch.cat(self.preds), group)
        t = gather_rows(torch.cat(self.target), group)
        _, _, per_class = ops.average_precision(p.float() if p.dtype != torch.float32 else p, t)
        return list(per_class.unbind(0))

    def reset(self) -> None:
        self.preds, self.target = [], []


class CosineSimilarity:
    """``nn.CosineSimilarity(dim=1)`` for logging (frame_transformer.py:121,257); no gradient."""

    def __init__(self, dim: int = 1, eps: float = 1e-8):
        if dim != 1:
            raise ValueError("only dim=1 (rows of a [B, C] pair) is used by the reference")
        self.eps = eps

    def __call__(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return ops.cosine_rows(a, b, self.eps)
