"""CPU restatement of the build-defined pyramid / cross-modal wiring (BASELINE configs[2]/[3]).

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

The stages are the reference's own operators, each already pinned elsewhere: the ResNet pyramid
(oracle.cnn_path.resnet_pyramid, pinned by tests/golden/resnet18_pyramid.npz from custom_resnet.py), the
pre-norm transformer / CLS plumbing / heads (oracle.clip_path, pinned by the vit.py goldens), BCE and the
hard-label distillation CE (frame_transformer.py:246-252).  The wiring itself has no reference class, so
this file is "parity unpinned" as a whole: it states what the HIP path must reproduce.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as TF

from . import clip_path as O
from . import cnn_path as C

Tensor = torch.Tensor


def lateral_tokens(x2: Tensor, x3: Tensor, x4: Tensor, P: Dict[str, Tensor]) -> Tensor:
    """x2 [F,128,2s,2s], x3 [F,256,s,s], x4 [F,512,s/2,s/2] -> [F, s*s, d]; GEMM-layout lateral weights:
    lat2.weight [d, (ki,kj,c)] = 2x2/2 convolution, lat4.weight [(ki,kj,co), 512] = 2x2/2 transposed convolution."""
    Fr, c2 = x2.shape[:2]
    d = P["lat3.weight"].shape[0]
    w2 = P["lat2.weight"].reshape(d, 2, 2, c2).permute(0, 3, 1, 2)                 # [d, c, ki, kj]
    t2 = TF.conv2d(x2, w2, P["lat2.bias"], stride=2)
    t3 = TF.conv2d(x3, P["lat3.weight"][:, :, None, None], P["lat3.bias"])
    c4 = x4.shape[1]
    w4 = P["lat4.weight"].reshape(2, 2, d, c4).permute(3, 2, 0, 1)                 # [cin, cout, ki, kj]
    t4 = TF.conv_transpose2d(x4, w4, None, stride=2)
    y = t2 + t3 + t4                                                                # [F, d, s, s]
    return y.flatten(2).transpose(1, 2)


def cross_attention(x: Tensor, c: Tensor, P: Dict[str, Tensor], prefix: str, heads: int) -> Tensor:
    q = O.linear(O.layernorm(x, P[prefix + "norm_q.weight"], P[prefix + "norm_q.bias"]), P[prefix + "to_q.weight"])
    kv = O.linear(O.layernorm(c, P[prefix + "norm_kv.weight"], P[prefix + "norm_kv.bias"]), P[prefix + "to_kv.weight"])
    B, Lq, inner = q.shape
    Lk, dh = c.shape[1], inner // heads
    q = q.view(B, Lq, heads, dh).transpose(1, 2)
    k, v = kv.view(B, Lk, 2, heads, dh).unbind(2)
    o = O.attention_core(q, k.transpose(1, 2), v.transpose(1, 2), dh ** -0.5)      # [B, H, Lq, dh]
    o = o.transpose(1, 2).reshape(B, Lq, inner)
    return x + O.linear(o, P[prefix + "to_out.weight"], P[prefix + "to_out.bias"])


def pyramid_vivit_forward(x: Tensor, audio: Optional[Tensor], P: Dict[str, Tensor], *, depth: int, heads: int,
                          layers=(2, 2, 2, 2), training_bn: bool = True, distill: bool = False):
    """x [b,t,3,H,W] (+ audio [b,A,audio_dim]) -> student logits [, teacher logits]."""
    b, t = x.shape[:2]
    bb = {k[len("backbone."):]: v for k, v in P.items() if k.startswith("backbone.")}
    x2, x3, x4, _ = C.resnet_pyramid(x.reshape(b * t, *x.shape[2:]), bb, layers=list(layers), training=training_bn)
    emb = lateral_tokens(x2, x3, x4, P)                                             # [b*t, n, d]
    n, d = emb.shape[1], emb.shape[2]
    tok = torch.cat((P["space_token"].expand(b * t, 1, d), emb), dim=1).view(b, t, n + 1, d)
    tok = (tok + P["pos_embedding"][:, :, : n + 1]).view(b * t, n + 1, d)
    s = O.prenorm_transformer(tok, P, "space_transformer.", depth, heads)
    seq = torch.cat((P["temporal_token"].expand(b, 1, d), s[:, 0].view(b, t, d)), dim=1)
    if audio is not None:
        a = O.linear(audio, P["audio_proj.weight"], P["audio_proj.bias"])
        seq = cross_attention(seq, a, P, "cross.", heads)
        if distill:
            seq = torch.cat((seq, a.mean(dim=1, keepdim=True)), dim=1)
    z = O.prenorm_transformer(seq, P, "temporal_transformer.", depth, heads)
    student = O.linear(O.layernorm(z[:, 0], P["mlp_head.0.weight"], P["mlp_head.0.bias"]), P["mlp_head.1.weight"],
                       P["mlp_head.1.bias"])
    if not distill:
        return student
    teacher = O.linear(O.layernorm(z[:, -1], P["distill_head.0.weight"], P["distill_head.0.bias"]),
                       P["distill_head.1.weight"], P["distill_head.1.bias"])
    return student, teacher
