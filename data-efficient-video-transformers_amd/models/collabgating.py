"""Mirror of the reference's ``src/models/collabgating.py`` on HIP kernels.

``CollaborativeGating().forward(batch)``: ``batch`` is the reference's nested list -- samples x scenes x experts, each
expert a ``[1, d_e]`` feature tensor (collabgating.py:18-57).  For every expert i of a scene:
``cur = projection(e_i)``; ``t_i = sum_j (cur + projection(other_j))`` over the other experts; ``attention =
projection(t_i)``; ``gated_i = F.glu(cat(cur, cur + attention))`` (ContextGating, :83-86); the scene vector is
``F.normalize(fc(sum_i gated_i))`` (GatedEmbeddingUnit, :62-71).  Result ``[B, scenes, 1024]``.

Reproduced quirks of the executable text: experts narrower than 2048 are stretched with nearest-neighbour
``F.interpolate`` (:11-15); an expert that has been visited is put back *projected* (:48), so experts visited
earlier enter later sums through a second projection: ``others_i = {proj(e_j): j > i} + {proj(proj(e_j)): j < i}``.
The file has no imports in the reference (NameError); the golden vectors come from executing it with the
missing names supplied (tools/gen_golden.py).  All scenes of the batch are evaluated together: one GEMM per
projection instead of one per expert pair.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as F
from ..lightning_compat import LightningModule


class GatedEmbeddingUnit(nn.Module):
    def __init__(self, input_dimension, output_dimension, use_bn):
        super().__init__()
        self.fc = nn.Linear(input_dimension, output_dimension)

    def forward(self, x):
        return F.l2_normalize(F.linear(x, self.fc.weight, self.fc.bias))


class ContextGating(nn.Module):
    def __init__(self, dimension, add_batch_norm=True):
        super().__init__()

    def forward(self, x, x1):
        return F.gate(x, F.add(x, x1))


class CollaborativeGating(LightningModule):
    def __init__(self, *, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.proj_input = 2048
        self.proj_embedding_size = 2048
        self.projection = nn.Linear(self.proj_input, self.proj_embedding_size)
        self.cg = ContextGating(self.proj_input)
        self.geu = GatedEmbeddingUnit(self.proj_input, 1024, False)
        self.compute_dtype = compute_dtype

    def pad(self, tensor):
        """nearest-neighbour stretch of [rows, d] to [rows, 2048] (F.interpolate default mode, :11-15)."""
        d = tensor.shape[-1]
        idx = (torch.arange(self.proj_input, device=tensor.device, dtype=torch.float32) * (d / self.proj_input)).floor().long()
        return tensor.index_select(-1, idx.clamp_(max=d - 1))

    def _stack(self, batch):
        """nested lists -> E tensors [B * S, 2048] (expert-major), B, S."""
        B, S, E = len(batch), len(batch[0]), len(batch[0][0])
        cols = [[] for _ in range(E)]
        for scenes in batch:
            if len(scenes) != S:
                raise ValueError("every sample must hold the same number of scenes")
            for experts in scenes:
                if len(experts) != E:
                    raise ValueError("every scene must hold the same number of experts")
                for e, t in enumerate(experts):
                    t = t.reshape(1, -1)
                    cols[e].append(t if t.shape[1] == self.proj_input else self.pad(t))
        return [F.cast(torch.cat(c, dim=0).contiguous(), self.compute_dtype) for c in cols], B, S

    def forward(self, batch):
        xs, B, S = self._stack(batch)
        E = len(xs)
        if E < 2:
            raise ValueError("collaborative gating needs at least two experts per scene (torch.stack of an empty "
                             "list in the reference, collabgating.py:40)")
        w, b = self.projection.weight, self.projection.bias
        p1 = [F.linear(x, w, b) for x in xs]                      # proj(e_j)
        p2 = [F.linear(p, w, b) for p in p1[:-1]]                 # proj(proj(e_j)) for experts already visited
        total = None
        for i in range(E):
            others = [p2[j] for j in range(i)] + [p1[j] for j in range(i + 1, E)]
            t = None
            for o in others:                                      # sum_j (cur + other_j)
                term = F.add(p1[i], o)
                t = term if t is None else F.add(t, term)
            gated = self.cg(p1[i], F.linear(t, w, b))
            total = gated if total is None else F.add(total, gated)
        out = self.geu(total)                                     # [B * S, 1024]
        return out.view(B, S, out.shape[-1])
