"""Mirror of the reference's ``src/models/losses/ntxent.py`` on HIP kernels.

``ContrastiveLoss(batch_size, temperature=0.5).forward(emb_i, emb_j)`` (ntxent.py:44-75): cosine-similarity matrix
of ``cat(emb_i, emb_j)``, positives on the +-batch_size diagonals, every other off-diagonal entry a negative,
``sum_k -log(exp(pos_k / T) / sum_{j != k} exp(sim_kj / T)) / (2 * batch_size)``.

``NT_Xent(batch_size, temperature, world_size)`` (ntxent.py:5-41) is the same objective written with
``CrossEntropyLoss(reduction="sum") / N``.  Deviations: the reference's ``forward`` computes the loss and returns
``None`` (no ``return``, :41) -- this one returns it; its mask construction (:19-22) pairs row ``i`` with
``batch_size + i`` and is therefore only consistent for ``world_size == 1`` -- here ``world_size > 1`` means
"the embeddings of all ranks": the local embeddings are all-gathered (rank order), the global loss is
evaluated on every rank and the backward pass returns the gradient of that global loss w.r.t. the *local*
embeddings.  Because every rank already holds the full loss, use a loss scale of 1 (not 1/world) for this term
under data parallelism: the all-reduced parameter gradient is then exactly the global one.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from ... import functional as F


class ContrastiveLoss(nn.Module):
    def __init__(self, batch_size, temperature=0.5):
        super().__init__()
        self.batch_size = batch_size
        self.register_buffer("temperature", torch.tensor(temperature))
        self.register_buffer("negatives_mask", (~torch.eye(batch_size * 2, batch_size * 2, dtype=bool)).float())
        self._t = float(temperature)

    def forward(self, emb_i, emb_j):
        if emb_i.shape != emb_j.shape or emb_i.shape[0] != self.batch_size:
            raise ValueError(f"expected two [{self.batch_size}, D] embedding batches")
        return F.contrastive_loss(emb_i, emb_j, self._t)


class _GatherRows(torch.autograd.Function):
    """all-gather along dim 0 (equal row counts); backward keeps this rank's slice of the gradient."""

    @staticmethod
    def forward(ctx, x, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous(), group=group)
        ctx.rows, ctx.rank = x.shape[0], rank
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, dy):
        return dy[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].contiguous(), None


class NT_Xent(nn.Module):
    def __init__(self, batch_size, temperature, world_size, group=None):
        super().__init__()
        self.batch_size, self.temperature, self.world_size, self.group = batch_size, float(temperature), world_size, group

    def forward(self, z_i, z_j):
        if self.world_size > 1:
            if not dist.is_initialized() or dist.get_world_size(self.group) != self.world_size:
                raise RuntimeError("NT_Xent(world_size > 1) needs an initialised process group of that size")
            z_i, z_j = _GatherRows.apply(z_i, self.group), _GatherRows.apply(z_j, self.group)
        if z_i.shape[0] != self.batch_size * self.world_size:
            raise ValueError(f"expected {self.batch_size * self.world_size} rows after the gather")
        return F.contrastive_loss(z_i, z_j, self.temperature)
