"""CPU oracle (TEST INFRASTRUCTURE, never shipped on the product path).

A plain-torch, explicit-formula restatement of the reference's video-clip
forward path.  Every function cites the reference lines it follows
(paths relative to /root/reference).  Arithmetic is written out (mean / var,
erf, matmul + softmax) instead of calling the fused ``torch.nn`` layers the
reference delegates to, so that the oracle is an independent statement of the
algorithm; ``tests/test_oracle_golden.py`` pins it against vectors generated
from the imported reference itself.

All functions are dtype-preserving: feed float64 tensors for a high-precision
check, float32 for the "reference CPU path" that bench.py times.  Backward
results come from torch autograd over these formulas.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# primitive operators
# --------------------------------------------------------------------------
def layernorm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """Row LayerNorm over the last dim, biased variance, affine.

    Reference: ``nn.LayerNorm(dim)`` in ``PreNorm`` src/models/vit.py:8-14 and the
    final norms at vit.py:64,75,105 (torch default eps 1e-5).
    """
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * weight + bias


def gelu_erf(x: Tensor) -> Tensor:
    """Exact (erf) GELU.  Reference: ``nn.GELU()`` src/models/vit.py:22,
    src/models/frame_transformer.py:106."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """y = x @ W^T (+ b), W stored [out, in] (``nn.Linear`` layout)."""
    y = x @ weight.transpose(-1, -2)
    return y if bias is None else y + bias


def softmax_lastdim(s: Tensor) -> Tensor:
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m)
    return e / e.sum(dim=-1, keepdim=True)


def attention_core(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    """softmax(q k^T * scale) v for [..., L, dh] operands.

    Reference: src/models/vit.py:51-55 (``dots = einsum(q,k) * scale``,
    ``softmax(dim=-1)``, ``einsum(attn, v)``).  Lq and Lk may differ, which is
    the cross-modal form used by SURVEY section 8 row a15.
    """
    s = (q @ k.transpose(-1, -2)) * scale
    return softmax_lastdim(s) @ v


def split_heads(t: Tensor, heads: int) -> Tensor:
    """'b n (h d) -> b h n d'  (src/models/vit.py:49)."""
    b, n, hd = t.shape
    return t.reshape(b, n, heads, hd // heads).permute(0, 2, 1, 3)


def merge_heads(t: Tensor) -> Tensor:
    """'b h n d -> b n (h d)'  (src/models/vit.py:56)."""
    b, h, n, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, n, h * d)


def self_attention(x: Tensor, w_qkv: Tensor, w_out: Optional[Tensor],
                   b_out: Optional[Tensor], heads: int) -> Tensor:
    """``Attention.forward`` src/models/vit.py:46-58.

    to_qkv has no bias (vit.py:39); q,k,v are the three chunks of the last dim
    (vit.py:48); scale = dim_head ** -0.5 (vit.py:37); ``to_out`` is
    Linear+Dropout unless heads == 1 and dim_head == dim (vit.py:34,41-44), in
    which case pass ``w_out=None``.
    """
    qkv = linear(x, w_qkv)
    inner = qkv.shape[-1] // 3
    q, k, v = qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:]
    dh = inner // heads
    o = attention_core(split_heads(q, heads), split_heads(k, heads),
                       split_heads(v, heads), dh ** -0.5)
    o = merge_heads(o)
    if w_out is None:
        return o
    return linear(o, w_out, b_out)


def feedforward(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """``FeedForward`` src/models/vit.py:17-28 (dropout p=0 / eval)."""
    return linear(gelu_erf(linear(x, w1, b1)), w2, b2)


def prenorm_transformer(x: Tensor, P: Dict[str, Tensor], prefix: str, depth: int,
                        heads: int) -> Tensor:
    """``Transformer.forward`` src/models/vit.py:71-75: per layer
    ``x = attn(LN(x)) + x; x = ff(LN(x)) + x``; final LayerNorm.

    ``P`` uses the reference's state-dict keys under ``prefix``
    (``layers.{i}.0.norm.weight`` ... ``norm.bias``)."""
    for i in range(depth):
        a = f"{prefix}layers.{i}.0."
        f = f"{prefix}layers.{i}.1."
        w_out = P.get(a + "fn.to_out.0.weight")
        b_out = P.get(a + "fn.to_out.0.bias")
        x = self_attention(layernorm(x, P[a + "norm.weight"], P[a + "norm.bias"]),
                           P[a + "fn.to_qkv.weight"], w_out, b_out, heads) + x
        x = feedforward(layernorm(x, P[f + "norm.weight"], P[f + "norm.bias"]),
                        P[f + "fn.net.0.weight"], P[f + "fn.net.0.bias"],
                        P[f + "fn.net.3.weight"], P[f + "fn.net.3.bias"]) + x
    return layernorm(x, P[prefix + "norm.weight"], P[prefix + "norm.bias"])


# --------------------------------------------------------------------------
# ViViT (factorised space -> time encoder over [b, t, c, H, W])
# --------------------------------------------------------------------------
def patchify(x: Tensor, patch: int) -> Tensor:
    """'b t c (h p1) (w p2) -> b t (h w) (p1 p2 c)'  src/models/vit.py:90.

    Patch-vector index = (p1 * P + p2) * C + c, i.e. channel fastest."""
    b, t, c, H, W = x.shape
    h, w = H // patch, W // patch
    x = x.reshape(b, t, c, h, patch, w, patch)
    x = x.permute(0, 1, 3, 5, 4, 6, 2)            # b t h w p1 p2 c
    return x.reshape(b, t, h * w, patch * patch * c)


def vivit_tokens(x: Tensor, P: Dict[str, Tensor], patch: int) -> Tensor:
    """Patch embedding + space CLS + learned positional table.
    src/models/vit.py:110-116 (emb_dropout = 0)."""
    e = linear(patchify(x, patch), P["to_patch_embedding.1.weight"],
               P["to_patch_embedding.1.bias"])
    b, t, n, d = e.shape
    cls = P["space_token"].reshape(1, 1, 1, d).expand(b, t, 1, d)
    tok = torch.cat((cls, e), dim=2)
    return tok + P["pos_embedding"][:, :, : n + 1]


def vivit_forward(x: Tensor, P: Dict[str, Tensor], *, patch: int, depth: int,
                  heads: int, pool: str = "cls") -> Tensor:
    """``ViViT.forward`` src/models/vit.py:109-128."""
    tok = vivit_tokens(x, P, patch)
    b, t, n1, d = tok.shape
    s = prenorm_transformer(tok.reshape(b * t, n1, d), P, "space_transformer.",
                            depth, heads)
    frame_cls = s[:, 0].reshape(b, t, d)                              # vit.py:120
    tcls = P["temporal_token"].reshape(1, 1, d).expand(b, 1, d)       # vit.py:122
    z = prenorm_transformer(torch.cat((tcls, frame_cls), dim=1), P,
                            "temporal_transformer.", depth, heads)   # vit.py:123-125
    pooled = z.mean(dim=1) if pool == "mean" else z[:, 0]            # vit.py:126
    h = layernorm(pooled, P["mlp_head.0.weight"], P["mlp_head.0.bias"])
    return linear(h, P["mlp_head.1.weight"], P["mlp_head.1.bias"])    # vit.py:104-107,128


# --------------------------------------------------------------------------
# losses (src/models/frame_transformer.py:89-90,246-273)
# --------------------------------------------------------------------------
def bce_with_logits(z: Tensor, y: Tensor) -> Tensor:
    """Mean over all elements of  max(z,0) - z*y + log(1+exp(-|z|)).
    ``nn.BCEWithLogitsLoss()`` frame_transformer.py:89,263,268,273."""
    return (torch.clamp(z, min=0) - z * y + torch.log1p(torch.exp(-z.abs()))).mean()


def cross_entropy_hard(student: Tensor, teacher: Tensor) -> Tensor:
    """Distillation term: CE(student, argmax(teacher)) frame_transformer.py:250."""
    idx = teacher.argmax(dim=-1)
    m = student.max(dim=-1, keepdim=True).values
    lse = m.squeeze(-1) + torch.log(torch.exp(student - m).sum(dim=-1))
    return (lse - student.gather(-1, idx[:, None]).squeeze(-1)).mean()


# --------------------------------------------------------------------------
# FrameTransformer token path (src/models/frame_transformer.py)
# --------------------------------------------------------------------------
def sinusoid_table(d_model: int, max_len: int, dtype=torch.float32) -> Tensor:
    """``PositionalEncoding`` buffer, frame_transformer.py:23-30.  NOTE the
    base is **1000**, not 10000 (``-math.log(1000.0) / d_model``, line 26).
    Returns [max_len, 1, d_model]."""
    pos = torch.arange(0, max_len, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32)
                    * (-math.log(1000.0) / d_model))
    pe = torch.zeros(max_len, d_model, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe[:, None, :].to(dtype)


def mha_seq_first(x: Tensor, kv: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor,
                  out_b: Tensor, nhead: int) -> Tensor:
    """torch ``nn.MultiheadAttention`` arithmetic (packed in_proj with bias,
    seq-first [L, B, E]) as used by ``TransformerEncoderLayer`` in
    frame_transformer.py:41-44.  ``kv`` = ``x`` for self-attention; a different
    [Lk, B, E] tensor gives the cross-modal form."""
    L, B, E = x.shape
    Lk = kv.shape[0]
    dh = E // nhead
    q = linear(x, in_w[:E], in_b[:E])
    k = linear(kv, in_w[E:2 * E], in_b[E:2 * E])
    v = linear(kv, in_w[2 * E:], in_b[2 * E:])
    q = q.reshape(L, B, nhead, dh).permute(1, 2, 0, 3)
    k = k.reshape(Lk, B, nhead, dh).permute(1, 2, 0, 3)
    v = v.reshape(Lk, B, nhead, dh).permute(1, 2, 0, 3)
    o = attention_core(q, k, v, dh ** -0.5)                # [B, h, L, dh]
    o = o.permute(2, 0, 1, 3).reshape(L, B, E)
    return linear(o, out_w, out_b)


def encoder_layer_postnorm(x: Tensor, P: Dict[str, Tensor], prefix: str, nhead: int) -> Tensor:
    """torch ``TransformerEncoderLayer`` defaults (post-norm, ReLU, eps 1e-5),
    eval mode: x = LN1(x + SA(x)); x = LN2(x + W2 relu(W1 x)).
    Instantiated at frame_transformer.py:41-44,99."""
    a = mha_seq_first(x, x, P[prefix + "self_attn.in_proj_weight"],
                      P[prefix + "self_attn.in_proj_bias"],
                      P[prefix + "self_attn.out_proj.weight"],
                      P[prefix + "self_attn.out_proj.bias"], nhead)
    x = layernorm(x + a, P[prefix + "norm1.weight"], P[prefix + "norm1.bias"])
    f = linear(torch.relu(linear(x, P[prefix + "linear1.weight"], P[prefix + "linear1.bias"])),
               P[prefix + "linear2.weight"], P[prefix + "linear2.bias"])
    return layernorm(x + f, P[prefix + "norm2.weight"], P[prefix + "norm2.bias"])


def transformer_base(x: Tensor, P: Dict[str, Tensor], prefix: str, nlayers: int,
                     nhead: int) -> Tensor:
    """``TransformerBase.forward`` frame_transformer.py:37-47 (no final norm)."""
    for i in range(nlayers):
        x = encoder_layer_postnorm(x, P, f"{prefix}transformer.layers.{i}.", nhead)
    return x


def mlp_head3(x: Tensor, P: Dict[str, Tensor], prefix: str = "img_mlp_head.") -> Tensor:
    """Linear-GELU-Linear-GELU-Linear, frame_transformer.py:106."""
    x = gelu_erf(linear(x, P[prefix + "0.weight"], P[prefix + "0.bias"]))
    x = gelu_erf(linear(x, P[prefix + "2.weight"], P[prefix + "2.bias"]))
    return linear(x, P[prefix + "4.weight"], P[prefix + "4.bias"])


# --------------------------------------------------------------------------
# Temporal pyramid (src/models/TPN.py:64-112)
# --------------------------------------------------------------------------
def sum_group(x: Tensor, groups: int) -> Tensor:
    """TPN.py:64-72: sum each run of ``groups`` consecutive frame vectors and
    concatenate the floor(pics/groups) sums along the feature dim."""
    b, pics, d = x.shape
    g = pics // groups
    return x[:, : g * groups].reshape(b, g, groups, d).sum(dim=2).reshape(b, g * d)


# --------------------------------------------------------------------------
# helpers for the CPU baseline leg of bench.py
# --------------------------------------------------------------------------
def vivit_step_fwd_bwd(x: Tensor, target: Tensor, P: Dict[str, Tensor], *, patch: int,
                       depth: int, heads: int) -> Tuple[Tensor, Dict[str, Tensor]]:
    """One forward + BCE loss + backward of the oracle ViViT.  Returns
    (loss, grads-by-key)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    logits = vivit_forward(x, leaves, patch=patch, depth=depth, heads=heads)
    loss = bce_with_logits(logits, target)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    return loss.detach(), {k: g for k, g in zip(leaves.keys(), grads)}
