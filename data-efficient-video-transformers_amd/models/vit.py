"""Drop-in mirror of the reference's ``src/models/vit.py`` on MI355X kernels.

Same class names, constructor signatures, ``forward`` signatures, submodule
layout and therefore the same ``state_dict`` keys as the reference
(``pos_embedding, space_token, temporal_token, to_patch_embedding.1.*,
{space,temporal}_transformer.layers.{i}.{0,1}.{norm,fn...}``, ``mlp_head.{0,1}.*``;
reference: src/models/vit.py:8-128), so reference checkpoints load unchanged.
``nn.Linear`` / ``nn.LayerNorm`` instances are parameter containers only: every
forward/backward runs through the HIP kernels of libdvt_hip.so (functional.py);
there is no torch arithmetic and no CPU path.

Build-side additions (keyword-only, defaults keep reference behaviour):
``compute_dtype`` -- activation / GEMM dtype (torch.bfloat16 default, torch.float32
for the fp32 parity mode); ``activation_checkpointing`` -- keep one activation per
transformer block and recompute the block in backward (BASELINE configs[4]).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as F


class Patchify(nn.Module):
    """Position 0 of ``to_patch_embedding`` (the reference's einops ``Rearrange``,
    vit.py:90).  Parameter-free; the gather itself is fused into ``F.patch_embed``."""

    def __init__(self, patch_size: int):
        super().__init__()
        self.patch_size = patch_size

    def forward(self, x):  # only reached when used stand-alone
        from .. import ops
        b, t = x.shape[0], x.shape[1]
        out = ops.patchify(x, self.patch_size, x.dtype)
        return out.view(b, t, -1, out.shape[-1])


class GELU(nn.Module):
    """Exact-erf GELU (vit.py:22); fused into the GEMM epilogue inside FeedForward."""

    def forward(self, x):
        return F.gelu(x)


class PreNorm(nn.Module):
    """vit.py:8-14."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        if isinstance(self.fn, (Attention, FeedForward)) and not kwargs:
            return self.fn(x, _norm=self.norm)
        return self.fn(F.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps), **kwargs)


class FeedForward(nn.Module):
    """vit.py:17-28: Linear(dim, hidden) - GELU - Dropout - Linear(hidden, dim) - Dropout."""

    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(dim, hidden_dim),
            GELU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, dim),
            nn.Dropout(dropout),
        )
        self.dropout_p = dropout

    def forward(self, x, _norm=None, _residual=False, _cdt=None):
        # _cdt: GEMM operand type when x is an fp32 residual stream (the launch-bound temporal stack, see ViViT.forward)
        l1, l2 = self.net[0], self.net[3]
        if self.training and self.dropout_p > 0.0:         # unfused: Linear - GELU - Dropout - Linear - Dropout (:20-26)
            h = F.layernorm(x, _norm.weight, _norm.bias, _norm.eps) if _norm is not None else x
            h = F.dropout(F.gelu(F.linear(h, l1.weight, l1.bias)), self.dropout_p, True)
            y = F.dropout(F.linear(h, l2.weight, l2.bias), self.dropout_p, True)
            return F.add(x, y) if _residual else y
        return F.mlp_block(x, _norm.weight if _norm is not None else None,
                           _norm.bias if _norm is not None else None,
                           l1.weight, l1.bias, l2.weight, l2.bias, act="gelu",
                           prenorm=_norm is not None, residual=_residual,
                           eps=_norm.eps if _norm is not None else 1e-5, cdt=_cdt)


class Attention(nn.Module):
    """vit.py:30-58."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(
            nn.Linear(inner_dim, dim),
            nn.Dropout(dropout),
        ) if project_out else nn.Identity()
        self.project_out = project_out
        self.dropout_p = dropout

    def forward(self, x, _norm=None, _residual=False, _cdt=None):
        w_out = self.to_out[0].weight if self.project_out else None
        b_out = self.to_out[0].bias if self.project_out else None
        if self.training and self.dropout_p > 0.0 and self.project_out:     # to_out = Linear + Dropout (:41-44)
            y = F.attn_block(x, _norm.weight if _norm is not None else None, _norm.bias if _norm is not None else None,
                             self.to_qkv.weight, w_out, b_out, self.heads, prenorm=_norm is not None, residual=False,
                             eps=_norm.eps if _norm is not None else 1e-5)
            y = F.dropout(y, self.dropout_p, True)
            return F.add(x, y) if _residual else y
        return F.attn_block(x, _norm.weight if _norm is not None else None,
                            _norm.bias if _norm is not None else None,
                            self.to_qkv.weight, w_out, b_out, self.heads,
                            prenorm=_norm is not None, residual=_residual,
                            eps=_norm.eps if _norm is not None else 1e-5, cdt=_cdt)


class Transformer(nn.Module):
    """vit.py:60-75: depth x [x = attn(LN x) + x; x = ff(LN x) + x], final LayerNorm."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.norm = nn.LayerNorm(dim)
        self.checkpoint = False          # build-side: recompute each block in backward (saves ~11 of 12 activations)
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout)),
            ]))

    def stream_f32_ok(self, rows: int, cdt: torch.dtype) -> bool:
        """May this stack run on an fp32 residual stream with ``cdt`` GEMM operands for ``rows`` rows?  (The launch-bound
        zone: 16-bit kernels, no active dropout, and every residual GEMM of a block a launch-bound shape -- the fp32
        residual epilogue is the panel-streaming kernel's.)"""
        if cdt not in (torch.bfloat16, torch.float16) or len(self.layers) == 0:
            return False
        attn, ff = self.layers[0][0].fn, self.layers[0][1].fn
        if not attn.project_out or (self.training and (attn.dropout_p > 0.0 or ff.dropout_p > 0.0)):
            return False
        d, inner, hidden = attn.to_out[0].weight.shape[0], attn.to_out[0].weight.shape[1], ff.net[0].weight.shape[0]
        from .. import ops
        return ops.gemm_is_launch_bound(rows, d, inner, cdt) and ops.gemm_is_launch_bound(rows, d, hidden, cdt)

    def forward_layers(self, x, cdt=None):
        """All residual blocks, without the final norm.  cdt: GEMM operand type when x is an fp32 stream."""
        for attn, ff in self.layers:
            if self.checkpoint and self.training and torch.is_grad_enabled():
                def block(t, attn=attn, ff=ff):
                    t = attn.fn(t, _norm=attn.norm, _residual=True, _cdt=cdt)
                    return ff.fn(t, _norm=ff.norm, _residual=True, _cdt=cdt)
                x = F.checkpoint(block, x, tuple(attn.parameters()) + tuple(ff.parameters()))
            else:
                x = attn.fn(x, _norm=attn.norm, _residual=True, _cdt=cdt)
                x = ff.fn(x, _norm=ff.norm, _residual=True, _cdt=cdt)
        return x

    def cls_prunable(self) -> bool:
        """True when the last layer may be evaluated for row 0 only (``forward_layers_cls``)."""
        if len(self.layers) == 0:
            return False
        attn = self.layers[-1][0].fn
        dropping = self.training and (attn.dropout_p > 0.0 or self.layers[-1][1].fn.dropout_p > 0.0)
        return attn.project_out and not dropping

    def forward_layers_cls(self, x, cdt=None):
        """``forward_layers(x)[:, 0]`` for x [S, N, d] -> [S, d]: what the reference reads of the space
        transformer (vit.py:119-120) and, under pool == 'cls', of the temporal one (:126).  In the last
        layer only the keys and values are computed for all rows; query, attention, output projection and
        the whole feed-forward run on row 0 of every sequence (F.attn_block_cls)."""
        *head, (attn, ff) = self.layers
        for a, f in head:
            if self.checkpoint and self.training and torch.is_grad_enabled():
                def block(t, a=a, f=f):
                    t = a.fn(t, _norm=a.norm, _residual=True, _cdt=cdt)
                    return f.fn(t, _norm=f.norm, _residual=True, _cdt=cdt)
                x = F.checkpoint(block, x, tuple(a.parameters()) + tuple(f.parameters()))
            else:
                x = a.fn(x, _norm=a.norm, _residual=True, _cdt=cdt)
                x = f.fn(x, _norm=f.norm, _residual=True, _cdt=cdt)

        def last(t):
            an, af = attn.norm, attn.fn
            c = F.attn_block_cls(t, an.weight, an.bias, af.to_qkv.weight, af.to_out[0].weight, af.to_out[0].bias,
                                 af.heads, eps=an.eps, cdt=cdt)
            return ff.fn(c, _norm=ff.norm, _residual=True, _cdt=cdt)

        if self.checkpoint and self.training and torch.is_grad_enabled():
            return F.checkpoint(last, x, tuple(attn.parameters()) + tuple(ff.parameters()))
        return last(x)

    def forward(self, x, cdt=None):
        x = self.forward_layers(x, cdt)
        return F.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)


class ViViT(nn.Module):
    """vit.py:79-128.  ``forward(x[b, t, c, H, W]) -> [b, num_classes]`` (fp32 logits)."""

    def __init__(self, image_size, patch_size, num_classes, num_frames, dim=192, depth=4, heads=3,
                 pool='cls', in_channels=3, dim_head=64, dropout=0., emb_dropout=0., scale_dim=4, *,
                 compute_dtype: torch.dtype = torch.bfloat16, activation_checkpointing: bool = False):
        super().__init__()
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        assert image_size % patch_size == 0, 'Image dimensions must be divisible by the patch size.'
        num_patches = (image_size // patch_size) ** 2
        patch_dim = in_channels * patch_size ** 2
        self.to_patch_embedding = nn.Sequential(
            Patchify(patch_size),
            nn.Linear(patch_dim, dim),
        )
        self.pos_embedding = nn.Parameter(torch.randn(1, num_frames, num_patches + 1, dim))
        self.space_token = nn.Parameter(torch.randn(1, 1, dim))
        self.space_transformer = Transformer(dim, depth, heads, dim_head, dim * scale_dim, dropout)
        self.temporal_token = nn.Parameter(torch.randn(1, 1, dim))
        self.temporal_transformer = Transformer(dim, depth, heads, dim_head, dim * scale_dim, dropout)
        self.dropout = nn.Dropout(emb_dropout)
        self.pool = pool
        self.mlp_head = nn.Sequential(
            nn.LayerNorm(dim),
            nn.Linear(dim, num_classes),
        )
        self.patch_size = patch_size
        self.num_patches = num_patches
        self.num_frames = num_frames
        self.emb_dropout_p = emb_dropout
        self.compute_dtype = compute_dtype
        self.zone_f32 = True          # fp32 residual stream in the temporal stack / heads under 16-bit kernels (see forward)
        self.space_transformer.checkpoint = activation_checkpointing
        self.temporal_transformer.checkpoint = activation_checkpointing

    def forward(self, x):
        pooled, pending = self._trunk(x)
        if pending is not None:                                                     # the temporal stack's final norm (:43)
            pooled = F.layernorm(pooled, pending.weight, pending.bias, pending.eps)
        hn, hl = self.mlp_head[0], self.mlp_head[1]
        h = F.layernorm(pooled, hn.weight, hn.bias, hn.eps)
        return F.linear(h, hl.weight, hl.bias, out_f32=True)                        # :128

    def loss(self, x, target):
        """nn.BCEWithLogitsLoss()(self(x), target) (the training step of the reference's Lightning wrapper), with the head
        -- final norm on the pooled row, mlp_head, the loss -- as one launch where the shape allows (8 rows at the metric
        shape: the separate kernels are pure launch cost).  -> (loss, logits)."""
        pooled, pending = self._trunk(x)
        hn, hl = self.mlp_head[0], self.mlp_head[1]
        if pooled.is_cuda and F.head_bce_supported(pooled, hl.weight):
            mixed = pooled.dtype == torch.float32 and self.compute_dtype in (torch.bfloat16, torch.float16)
            return F.head_bce(pooled, pending, hn, hl, target, lp_dtype=self.compute_dtype if mixed else None)
        if pending is not None:
            pooled = F.layernorm(pooled, pending.weight, pending.bias, pending.eps)
        logits = F.linear(F.layernorm(pooled, hn.weight, hn.bias, hn.eps), hl.weight, hl.bias, out_f32=True)
        return F.bce_with_logits(logits, target), logits

    def _trunk(self, x):
        """-> (pooled [b, dim], the LayerNorm still to be applied to it or None)."""
        if x.dim() != 5:
            raise ValueError("ViViT expects a clip tensor [b, t, c, H, W]")
        b, t = x.shape[0], x.shape[1]
        if t != self.num_frames:
            raise ValueError(f"clip has {t} frames but pos_embedding was built for {self.num_frames} "
                             "(vit.py:94,115 broadcast)")
        T = self.compute_dtype
        F.lp_clear()            # 16-bit gradient copies a previous step's backward left untaken (functional._lp_hand)
        n = (x.shape[3] // self.patch_size) * (x.shape[4] // self.patch_size)
        pe = self.to_patch_embedding[1]
        emb = F.patch_embed(x, pe.weight, pe.bias, self.patch_size, T)              # vit.py:110
        tok = F.tokens_assemble(emb, self.space_token, self.pos_embedding, b * t, t, n)   # :113-115
        tok = F.dropout(tok, self.emb_dropout_p, self.training)                            # :116
        st, tt = self.space_transformer, self.temporal_transformer
        sn = st.norm
        if st.cls_prunable():                                                       # only x[:, 0] is read (:120)
            s = st.forward_layers_cls(tok).view(b * t, 1, -1)
        else:
            s = st.forward_layers(tok)                                              # :118-119
        # The temporal stack and the heads are the launch-bound zone of the step (b (t + 1) rows: 264 at the metric shape):
        # single rows carry whole gradients there and nothing averages the rounding of 16-bit storage, while fp32 storage
        # of so few rows costs no bandwidth -- its residual stream and row gradients are kept in fp32 (GEMM operands stay T).
        zone32 = self.zone_f32 and tt.stream_f32_ok(b * (t + 1), T)
        cdt = T if zone32 else None
        seq = F.cls_norm_concat(s, sn.weight, sn.bias, self.temporal_token, b, t, sn.eps,
                                out_dtype=torch.float32 if zone32 else None)       # :119-123
        if self.pool == 'cls' and tt.cls_prunable():                                # only x[:, 0] is read (:126)
            return tt.forward_layers_cls(seq, cdt), tt.norm
        z = tt(seq, cdt)                                                            # :125
        return (F.mean_rows(z) if self.pool == 'mean' else F.select_first_row(z)), None  # :126
