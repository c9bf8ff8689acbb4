"""CPU restatement of the multi-modal gating and the contrastive objective.

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Pinned by tests/golden/fusion.npz: ``ContrastiveLoss`` from the imported reference
(src/models/losses/ntxent.py:44-75) and ``CollaborativeGating`` from the executed reference text
(src/models/collabgating.py:2-87; the file lacks its imports, tools/gen_golden.py supplies them).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import clip_path as O

Tensor = torch.Tensor


def cosine_similarity_matrix(z: Tensor, eps: float = 1e-8) -> Tensor:
    zn = z / z.norm(dim=1, keepdim=True).clamp_min(eps)
    return zn @ zn.t()


def contrastive_loss(z_i: Tensor, z_j: Tensor, temperature: float) -> Tensor:
    """ntxent.py:53-75."""
    B = z_i.shape[0]
    sim = cosine_similarity_matrix(torch.cat((z_i, z_j), dim=0))
    M = 2 * B
    pos = torch.cat((torch.diagonal(sim, B), torch.diagonal(sim, -B)))
    off = ~torch.eye(M, dtype=torch.bool)
    denom = (torch.exp(sim / temperature) * off).sum(dim=1)
    return (-torch.log(torch.exp(pos / temperature) / denom)).sum() / M


def stretch_nearest(t: Tensor, width: int) -> Tensor:
    """F.interpolate(t[None], width)[0] (nearest): out[:, i] = t[:, floor(i * d / width)] (collabgating.py:11-15)."""
    d = t.shape[-1]
    idx = (torch.arange(width, dtype=torch.float32) * (d / width)).floor().long().clamp(max=d - 1)
    return t[..., idx]


def collaborative_gating(batch: List[List[List[Tensor]]], P: Dict[str, Tensor]) -> Tensor:
    """collabgating.py:18-57 with its list mutation made explicit."""
    w, b = P["projection.weight"], P["projection.bias"]
    out = []
    for scenes in batch:
        rows = []
        for experts in scenes:
            xs = [e if e.shape[1] == 2048 else stretch_nearest(e, 2048) for e in experts]
            E = len(xs)
            p1 = [O.linear(x, w, b) for x in xs]
            p2 = [O.linear(p, w, b) for p in p1]
            total = 0
            for i in range(E):
                others = [p2[j] for j in range(i)] + [p1[j] for j in range(i + 1, E)]
                t = sum(p1[i] + o for o in others)
                att = O.linear(t, w, b)
                total = total + p1[i] * torch.sigmoid(p1[i] + att)
            v = O.linear(total, P["geu.fc.weight"], P["geu.fc.bias"])
            rows.append(v / v.norm(dim=1, keepdim=True).clamp_min(1e-12))
        out.append(torch.stack(rows))
    return torch.stack(out).squeeze(2)
