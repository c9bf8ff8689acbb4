"""BASELINE configs[2] ("spatial-temporal pyramid, 3 scales") and configs[3] ("multi-modal
cross-attention, video + 1-D audio tokens, distillation loss head") -- SURVEY section 8d configs 3/4.

The reference holds the *parts* (per-frame ``custom_resnet`` pyramid returning ``(x2, x3, x4)``,
custom_resnet.py:138-153; the factorised space -> time ``Transformer`` of vit.py:60-128; the
token-injection + CLS/"distillation token" read-out and the BCE + hard-label CE distillation loss of
frame_transformer.py:225-239,246-252) but no module that wires them at these shapes.  This module is
that wiring -- **build-defined**, not a mirror of one reference class; every stage is the mirrored
reference component or an operator already covered by the parity tests:

  frames [b*t, 3, H, W] -> resnet pyramid (x2 128 x 2s x 2s, x3 256 x s x s, x4 512 x s/2 x s/2, s = H/16)
  -> three lateral projections onto the middle grid, summed (FPN-style):
        lat2: 2x2 stride-2 convolution 128 -> d        (im2col + GEMM)
        lat3: 1x1 convolution          256 -> d        (GEMM)
        lat4: 2x2 stride-2 transposed convolution 512 -> d   (GEMM + pixel shuffle)
  -> n = s*s tokens of width d per frame (196 at 224^2: the metric-shape token count)
  -> space CLS + learned positions -> space Transformer -> per-frame CLS -> temporal sequence [b, t+1, d]
  -> (configs[3]) cross-modal block: the temporal sequence attends to ``audio_tokens`` audio tokens
     (Linear(audio_dim -> d)); queries video, keys/values audio, Lq != Lk
  -> (distill) the pooled audio embedding is appended as the last token (the reference's "distillation
     token", frame_transformer.py:225-226,233-236)
  -> temporal Transformer -> ``mlp_head(CLS)`` [, ``distill_head(last token)``].

Lateral weights are stored in GEMM layout: ``lat2.weight [d, (ki, kj, c)]``, ``lat4.weight [(ki, kj, co), 512]``.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as F
from . import custom_resnet
from .vit import Transformer


class CrossAttention(nn.Module):
    """x + to_out(softmax(q k^T / sqrt(dh)) v), q from LN(x), k/v from LN(context)."""

    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.norm_q = nn.LayerNorm(dim)
        self.norm_kv = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim)

    def forward(self, x, context):
        return F.cross_attn_block(x, context, self.norm_q.weight, self.norm_q.bias, self.norm_kv.weight,
                                  self.norm_kv.bias, self.to_q.weight, self.to_kv.weight, self.to_out.weight,
                                  self.to_out.bias, self.heads, self.norm_q.eps)


class PyramidViViT(nn.Module):
    def __init__(self, image_size=224, num_classes=19, num_frames=32, dim=512, depth=4, heads=8, dim_head=64,
                 scale_dim=4, backbone="resnet18", audio_tokens=0, audio_dim=128, distill=False, *,
                 compute_dtype: torch.dtype = torch.bfloat16, activation_checkpointing: bool = False):
        super().__init__()
        if image_size % 32:
            raise ValueError("image_size must be a multiple of 32 (x4 is the /32 map)")
        if distill and not audio_tokens:
            raise ValueError("the distillation token is the pooled audio embedding: audio_tokens must be > 0")
        self.backbone = getattr(custom_resnet, backbone)(False, compute_dtype=compute_dtype)
        exp = 4 if backbone in ("resnet50", "resnet101", "resnet152") else 1
        self.c2, self.c3, self.c4 = 128 * exp, 256 * exp, 512 * exp
        s = image_size // 16
        self.grid, self.num_patches, self.num_frames, self.dim = s, s * s, num_frames, dim
        self.lat2 = nn.Linear(4 * self.c2, dim)
        self.lat3 = nn.Linear(self.c3, dim)
        self.lat4 = nn.Linear(self.c4, 4 * dim, bias=False)
        self.pos_embedding = nn.Parameter(torch.randn(1, num_frames, self.num_patches + 1, dim))
        self.space_token = nn.Parameter(torch.randn(1, 1, dim))
        self.space_transformer = Transformer(dim, depth, heads, dim_head, dim * scale_dim)
        self.temporal_token = nn.Parameter(torch.randn(1, 1, dim))
        self.temporal_transformer = Transformer(dim, depth, heads, dim_head, dim * scale_dim)
        self.space_transformer.checkpoint = self.temporal_transformer.checkpoint = activation_checkpointing
        self.audio_tokens = audio_tokens
        if audio_tokens:
            self.audio_proj = nn.Linear(audio_dim, dim)
            self.cross = CrossAttention(dim, heads, dim_head)
        self.distill = distill
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))
        if distill:
            self.distill_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))
        self.compute_dtype = compute_dtype

    # ------------------------------------------------------------------ pyramid -> tokens
    def pyramid_tokens(self, frames):
        """frames [F, 3, H, W] -> [F * n, d]."""
        Fr = frames.shape[0]
        (x2, _, H2, W2), (x3, _, H3, W3), (x4, _, H4, W4) = self.backbone.forward_nhwc(frames)
        if (H2, W2) != (2 * H3, 2 * W3) or (H3, W3) != (2 * H4, 2 * W4):
            raise ValueError("pyramid levels must halve exactly (input side a multiple of 32)")
        t2 = F.linear(F.im2col_nhwc(x2, Fr, self.c2, H2, W2, 2, 2, 0), self.lat2.weight, self.lat2.bias)
        t3 = F.linear(x3, self.lat3.weight, self.lat3.bias)
        t4 = F.col2im_nhwc(F.linear(x4, self.lat4.weight), Fr, self.dim, H3, W3, 2, 2, 0)
        return F.add(F.add(t2, t3), t4)

    def forward(self, x, audio=None):
        if x.dim() != 5:
            raise ValueError("PyramidViViT expects a clip tensor [b, t, c, H, W]")
        b, t = x.shape[0], x.shape[1]
        if t != self.num_frames:
            raise ValueError(f"clip has {t} frames but pos_embedding was built for {self.num_frames}")
        if bool(self.audio_tokens) != (audio is not None):
            raise ValueError("audio tokens must be given exactly when the model was built with audio_tokens > 0")
        n, T = self.num_patches, self.compute_dtype
        emb = self.pyramid_tokens(x.reshape(b * t, *x.shape[2:]))
        tok = F.tokens_assemble(emb, self.space_token, self.pos_embedding, b * t, t, n)
        st = self.space_transformer
        sn = st.norm
        # only row 0 of the space transformer's output is read (vit.py:119-120): last layer on the CLS rows
        s = st.forward_layers_cls(tok).view(b * t, 1, -1) if st.cls_prunable() else st.forward_layers(tok)
        seq = F.cls_norm_concat(s, sn.weight, sn.bias, self.temporal_token, b, t, sn.eps)       # [b, t+1, d]
        if self.audio_tokens:
            if audio.shape[:2] != (b, self.audio_tokens):
                raise ValueError(f"audio must be [b, {self.audio_tokens}, audio_dim]")
            a = F.linear(F.cast(audio, T), self.audio_proj.weight, self.audio_proj.bias)        # [b, A, d]
            seq = self.cross(seq, a)
            if self.distill:                                  # append the pooled audio embedding as the last token
                inj = F.mean_rows(a).view(1, b, self.dim)
                seq = F.to_seq_first(F.concat_rows(F.to_seq_first(seq), inj))                   # [b, t+2, d]
        z = self.temporal_transformer(seq)
        zs = F.to_seq_first(z)                                                                  # [L, b, d]
        hn, hl = self.mlp_head[0], self.mlp_head[1]
        student = F.linear(F.layernorm(F.select_seq_first_row(zs, 0), hn.weight, hn.bias, hn.eps), hl.weight,
                           hl.bias, out_f32=True)
        if not self.distill:
            return student
        dn, dl = self.distill_head[0], self.distill_head[1]
        teacher = F.linear(F.layernorm(F.select_seq_first_row(zs, zs.shape[0] - 1), dn.weight, dn.bias, dn.eps),
                           dl.weight, dl.bias, out_f32=True)
        return student, teacher

    # ------------------------------------------------------------------ Lightning-style step (frame_transformer.py:246-252)
    def training_step(self, batch, batch_idx=0):
        if self.audio_tokens:
            target, clip, audio = batch
        else:
            (target, clip), audio = batch, None
        out = self(clip, audio)
        if self.distill:
            student, teacher = out
            base = F.bce_with_logits(student, target.float())
            dist = F.cross_entropy_argmax(student, teacher)               # hard-label distillation (:250)
            return F.add(base.reshape(1), dist.reshape(1)).reshape(())
        return F.bce_with_logits(out, target.float())
