"""Mirror of the reference's ``src/models/frame_transformer.py`` token path on HIP kernels.

Kept: class names, ``FrameTransformer(**config)``, ``forward(img, vid)``, ``vid_step``,
``img_step``, ``distillation_step``, ``training_step / validation_step / test_step``,
``configure_optimizers``, the attribute / state-dict names
(``position_encoder.pe``, ``distil_transformer.transformer.layers.{i}.self_attn.in_proj_weight``
..., ``vid_cls``, ``img_mlp_head.{0,2,4}``, ``norm``), the hyper-parameter mutation
``seq_len += 1`` when ``cls`` (frame_transformer.py:87-88) and the ``running_logits`` /
``running_labels`` lists read by the callbacks.

Built here (SURVEY section 8 rows a8, a9, a10 token part, a15/a16 token part, a17, a18):
sinusoidal positional encoding (base 1000), the post-norm ReLU ``TransformerBase``
(torch ``TransformerEncoderLayer`` arithmetic), the 3-layer GELU head, BCE + hard-label
distillation losses, the CLS-clip concatenation, and the cross-modal ``sum`` /
``distil`` injection (the video CLS embedding joins the image tokens and they
self-attend jointly).

CNN encoders: ``ImgResNet`` (ResNet-18 -> 896, frozen) and ``VidResNet`` (R(2+1)D-18 -> 896) are built on
the im2col + MFMA-GEMM + BatchNorm kernels (models/custom_resnet.py, models/video_resnet.py) with
torchvision's state-dict key layout; ``pretrained=True`` needs the network, so they start from random
weights unless a state dict is loaded, and the R(2+1)D block is parity-UNPINNED (torchvision is not
installed; checked against a torch-CPU conv3d restatement).  ``vid_encoder`` / ``img_encoder`` remain
injectable (any module mapping frames / chunks to ``d_model`` vectors; the small tests inject a patch-linear
stand-in of their own, tests/encoders.py).

Deviations from the literal reference text, all where the reference does not execute
(SURVEY section 8a notes): missing ``img_cls`` / ``img_model`` / ``scene_transformer``
members are created; ``torch.cat((data, distil_inject))`` gets the missing
``unsqueeze(0)``; the positional table is sized to the sequence actually used;
``distil`` returns 19-d logits for both streams; ``view(batch_size, ...)`` uses the
tensor's batch, not ``hparams.batch_size``; dropout p > 0 in training mode draws its masks from the Philox
dropout kernel (torch's generator stream cannot be reproduced; parity is defined in eval mode).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from .. import functional as F
from .. import optim
from ..lightning_compat import LightningModule


class PositionalEncoding(LightningModule):
    """frame_transformer.py:19-34 (duplicate transformer.py:10-25).  NOTE base **1000**
    (``-math.log(1000.0) / d_model``, line 26).  Buffer ``pe``: [max_len, 1, d_model]."""

    def __init__(self, d_model, dropout=0.1, max_len=4):
        super(PositionalEncoding, self).__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.p = dropout
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(1000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        pe = pe.unsqueeze(0).transpose(0, 1)
        self.register_buffer('pe', pe)

    def forward(self, x):
        return F.dropout(F.add_positional_table(x, self.pe), self.p, self.training)       # :33-34


class EncoderLayer(nn.Module):
    """Parameter container with torch ``nn.TransformerEncoderLayer`` names
    (``self_attn.in_proj_weight/bias``, ``self_attn.out_proj.*``, ``linear1/2``,
    ``norm1/2``): post-norm, ReLU, seq-first, eps 1e-5."""

    class _SelfAttn(nn.Module):
        def __init__(self, d, nhead):
            super().__init__()
            self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
            self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
            self.out_proj = nn.Linear(d, d)
            nn.init.xavier_uniform_(self.in_proj_weight)
            nn.init.constant_(self.out_proj.bias, 0.)
            self.num_heads = nhead

    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = EncoderLayer._SelfAttn(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout
        self.nhead = nhead

    def forward(self, x):
        """x [L, B, E]:  x = LN1(x + drop(SA(x)));  x = LN2(x + drop(W2 drop(relu(W1 x)))).
        Training-mode dropout (p > 0) runs the unfused composition with the Philox dropout kernel at torch's
        three hidden-state sites (dropout1, dropout, dropout2) and on the attention probabilities
        (``MultiheadAttention(dropout=p)``, inside the attention kernel)."""
        a = self.self_attn
        if self.training and self.p > 0.0:
            # (F.fork: a tensor with two consumers -- the sublayer and its residual add -- gets its two gradients summed by
            # dvt_add instead of by autograd's own accumulate kernel: no ATen kernel inside the step)
            x, xr = F.fork(x)
            sa = F.attn_block(x, None, None, a.in_proj_weight, a.out_proj.weight, a.out_proj.bias, self.nhead,
                              prenorm=False, residual=False, b_qkv=a.in_proj_bias, seq_first=True, attn_dropout=self.p)
            # (dropout fused with the residual add / with the ReLU in front of it: the same masks at the same RNG sites, 3
            #  launches fewer per layer forward and one fewer backward -- each ~4.7 us of a 28-row step)
            x = F.layernorm(F.dropout_add(sa, xr, self.p, True), self.norm1.weight, self.norm1.bias, self.norm1.eps)
            x, xr = F.fork(x)
            h = F.relu_dropout(F.linear(x, self.linear1.weight, self.linear1.bias), self.p, True)
            y = F.linear(h, self.linear2.weight, self.linear2.bias)
            return F.layernorm(F.dropout_add(y, xr, self.p, True), self.norm2.weight, self.norm2.bias, self.norm2.eps)
        x = F.attn_block(x, None, None, a.in_proj_weight, a.out_proj.weight, a.out_proj.bias, self.nhead,
                         prenorm=False, residual=True, b_qkv=a.in_proj_bias, seq_first=True)
        x = F.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = F.mlp_block(x, None, None, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                        self.linear2.bias, act="relu", prenorm=False, residual=True)
        return F.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)


class Encoder(nn.Module):
    """``nn.TransformerEncoder(layer, n)`` container: ``layers.{i}.*`` keys, no final norm."""

    def __init__(self, d_model, nhead, nhid, nlayers, dropout):
        super().__init__()
        self.layers = nn.ModuleList([EncoderLayer(d_model, nhead, nhid, dropout) for _ in range(nlayers)])

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class TransformerBase(LightningModule):
    """frame_transformer.py:37-47 (``output_dimension`` is unused there as well)."""

    def __init__(self, input_dimension, output_dimension, nhead, nhid, nlayers, dropout):
        super(TransformerBase, self).__init__()
        self.transformer = Encoder(input_dimension, nhead, nhid, nlayers, dropout)

    def forward(self, x):
        return self.transformer(x)


class ImgResNet(LightningModule):
    """frame_transformer.py:50-61: ResNet-18 backbone with ``fc -> Sequential(Linear(512, 896))``, run under
    ``torch.no_grad()`` (frozen).  ``pretrained=True`` of the reference needs the network; weights are
    random unless a state dict (torchvision resnet18 key layout) is loaded."""

    def __init__(self, compute_dtype=torch.bfloat16):
        super(ImgResNet, self).__init__()
        from .custom_resnet import resnet18
        self.backbone = resnet18(False, compute_dtype=compute_dtype)
        num_filters = self.backbone.fc.in_features
        self.backbone.fc = nn.Sequential(nn.Linear(num_filters, 896))

    def forward(self, x):
        with torch.no_grad():
            x4, N, H, W = self.backbone.forward_nhwc(x)[-1]
            pooled = F.mean_rows(x4.view(N, H * W, x4.shape[1]))            # avgpool + flatten
            fc = self.backbone.fc[0]
            return F.linear(pooled, fc.weight, fc.bias)


class VidResNet(LightningModule):
    """frame_transformer.py:64-74: R(2+1)D-18 backbone with ``fc -> Sequential(Linear(512, 896))``, trainable."""

    def __init__(self, compute_dtype=torch.bfloat16):
        super(VidResNet, self).__init__()
        from .video_resnet import r2plus1d_18
        self.backbone = r2plus1d_18(False, compute_dtype=compute_dtype)
        num_filters = self.backbone.fc.in_features
        self.backbone.fc = nn.Sequential(nn.Linear(num_filters, 896))

    def forward(self, x):
        return self.backbone(x)


class _MissingEncoder(nn.Module):
    def __init__(self, what):
        super().__init__()
        self.what = what

    def forward(self, x):
        raise NotImplementedError(f"{self.what}: pass vid_encoder= / img_encoder= to FrameTransformer")


class FrameTransformer(LightningModule):
    def __init__(self, **kwargs):
        super(FrameTransformer, self).__init__()
        self.save_hyperparameters()
        hp = self.hparams
        if hp.get("cls", 0):
            hp.seq_len = hp.get("seq_len", 13) + 1
        # constants hard-coded in the reference, promoted to keyword arguments (additive)
        d = hp.get("d_model", 896)
        drop = hp.get("encoder_dropout", 0.5)
        self.d_model = d
        self.tokens = hp.get("tokens", 14)
        self.frame_len = hp.get("frame_len", 12)
        self.clip_size = hp.get("clip_size", 112)
        self.img_size = hp.get("img_size", 224)
        self.n_out = hp.get("n_out", 19)
        self.compute_dtype = hp.get("compute_dtype", torch.bfloat16)
        from .. import metrics as _metrics
        self.train_aprc = _metrics.AveragePrecision(num_classes=self.n_out)     # :114
        self.val_aprc = _metrics.AveragePrecision(num_classes=self.n_out)       # :118
        self.cos = _metrics.CosineSimilarity(dim=1)                             # :121
        self.criterion = F.bce_with_logits                        # nn.BCEWithLogitsLoss()  :89
        self.distil_criterion = F.cross_entropy_argmax            # CE(student, argmax(teacher))  :90,250
        self.position_encoder = PositionalEncoding(d, drop, max_len=self.tokens + 1)      # :91-93 (+1: injected token)
        # default encoders = the reference's (random init: no pretrained download); injectable for other sizes.
        # The image branch (img_model / scene_transformer / img_cls: commented out in the reference, :94,98,104) is
        # only built for the modes that use it, so a "vid" model has exactly the reference's state-dict keys and loads
        # its checkpoints with strict=True.
        mode = hp.get("model", "vid")
        self.has_img_branch = mode in ("distil", "sum", "sum_residual", "post_sum", "frame", "pre_modal")
        if self.has_img_branch:
            self.img_model = hp.get("img_encoder", None) or (ImgResNet(self.compute_dtype) if d == 896 else
                                                             _MissingEncoder("img_encoder for d_model != 896"))   # :94
            self.scene_transformer = TransformerBase(d, d, hp.get("scene_nhead", 4), hp.get("scene_nhid", 896), 4, drop)  # :98
        self.vid_model = hp.get("vid_encoder", None) or (VidResNet(self.compute_dtype) if d == 896 else
                                                         _MissingEncoder("vid_encoder for d_model != 896"))   # :95
        self.distil_transformer = TransformerBase(d, 128, hp.get("vid_nhead", 2), hp.get("vid_nhid", 512), 4, drop)  # :99
        self.running_labels = []
        self.running_logits = []
        self.running_paths = []
        self.running_embeds = []
        if self.has_img_branch:
            self.img_cls = nn.Parameter(torch.rand(1, 3, self.img_size, self.img_size))                   # :104
        self.vid_cls = nn.Parameter(torch.rand(1, self.frame_len, 3, self.clip_size, self.clip_size))    # :105
        self.img_mlp_head = nn.Sequential(nn.Linear(d, 512), nn.GELU(), nn.Linear(512, 128), nn.GELU(),
                                          nn.Linear(128, self.n_out))                                     # :106
        self.norm = nn.LayerNorm(d)                                                                       # :117
        for k in ("img_encoder", "vid_encoder"):       # modules are attributes, not hyper-parameters
            if k in hp:
                delattr(hp, k)
        # In the image-only modes the video branch (always constructed by the reference) takes no part in the computation
        # and must not be touched by the optimizer either (torch skips parameters without a gradient; a flat-buffer
        # optimizer would still decay them), so it is frozen.  ``norm`` (:117) is unused by every forward path.
        idle = {"vid": (),
                "frame": ("vid_model", "distil_transformer", "vid_cls"),
                "pre_modal": ("vid_model", "distil_transformer", "vid_cls")}.get(hp.get("model", ""), ())
        for name in idle + ("norm",):
            m = getattr(self, name)
            for p in ([m] if isinstance(m, nn.Parameter) else m.parameters()):
                p.requires_grad_(False)

    # ------------------------------------------------------------------ optimizer (:123-134)
    def configure_optimizers(self):
        hp = self.hparams
        if hp.opt == "sgd":
            return optim.SGD(self.parameters(), lr=hp.learning_rate, momentum=hp.momentum,
                                   weight_decay=hp.weight_decay)
        if hp.opt == "adamW":
            return optim.AdamW(self.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
        if hp.opt == "adagrad":
            return optim.Adagrad(self.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
        raise ValueError(f"unknown optimizer {hp.opt!r}")

    # ------------------------------------------------------------------ heads
    def _head(self, x):
        h = self.img_mlp_head
        x = F.gelu(F.linear(x, h[0].weight, h[0].bias))
        x = F.gelu(F.linear(x, h[2].weight, h[2].bias))
        return F.linear(x, h[4].weight, h[4].bias, out_f32=True)

    def _with_cls(self, data, cls):
        """Per-sample cat of the learnable pixel-space CLS item (:194-197, :213-217):
        data [B, S, ...], cls [1, ...] -> [B, S+1, ...]."""
        B, S = data.shape[0], data.shape[1]
        out = F.cls_concat(data.reshape(B, S, -1), cls)
        return out.view((B, S + 1) + tuple(data.shape[2:]))

    # ------------------------------------------------------------------ vid_step (:192-210)
    def vid_step(self, data):
        B = data.shape[0]
        data = self._with_cls(data, self.vid_cls)                       # [B, 14, 12, 3, 112, 112]
        data = data.reshape(-1, self.frame_len, 3, self.clip_size, self.clip_size)
        data = data.permute(0, 2, 1, 3, 4)                              # [B*14, 3, 12, 112, 112]
        bb = getattr(self.vid_model, "backbone", None)
        if bb is not None and getattr(self, "restrict_pixel_grad", True):
            # only the CLS chunk (chunk 0 of every sample) carries a pixel gradient: tell the stem
            S1 = data.shape[0] // B
            bb.input_grad_clips = [b * S1 for b in range(B)]
        emb = self.vid_model(data)                                      # [B*14, 896]
        if bb is not None:
            bb.input_grad_clips = None
        if self.hparams.model == "pre-modal":
            return emb
        emb = F.cast(emb, self.compute_dtype).reshape(B, -1, self.d_model)
        seq = F.to_seq_first(emb)                                       # [14, B, 896]
        seq = self.position_encoder(seq)
        seq = self.distil_transformer(seq)
        return F.select_seq_first_row(seq, 0)                           # CLS -> [B, 896]

    # ------------------------------------------------------------------ img_step (:212-244)
    def img_step(self, data, distil_inject):
        B = data.shape[0]
        data = self._with_cls(data, self.img_cls)                       # [B, seq_len, 3, 224, 224]
        emb = self.img_model(data.reshape(-1, 3, self.img_size, self.img_size))
        emb = F.cast(emb, self.compute_dtype).reshape(B, -1, self.d_model)
        seq = F.to_seq_first(emb)                                       # [S, B, 896]
        mode = self.hparams.model
        if mode in ("sum", "distil", "post_sum") and distil_inject is not None:
            # cross-modal injection: the video CLS embedding becomes one more token (:225-226)
            seq = F.concat_rows(seq, F.cast(distil_inject, self.compute_dtype).unsqueeze(0))
        seq = self.position_encoder(seq)
        seq = self.scene_transformer(seq)
        cls = F.select_seq_first_row(seq, 0)
        if mode in ("distil", "sum", "post_sum"):
            return cls, F.select_seq_first_row(seq, seq.shape[0] - 1)   # (:233-239)
        if mode == "sum_residual":
            return cls, seq
        return self._head(cls)

    def distillation_step(self, img, vid):
        vid_cls = self.vid_step(vid)
        return self.img_step(img, vid_cls)

    # ------------------------------------------------------------------ forward (:136-190)
    def forward(self, img, vid):
        mode = self.hparams.model
        if mode == "distil":
            img_cls, vid_tkn = self.distillation_step(img, vid)
            return self._head(img_cls), self._head(vid_tkn)
        if mode == "sum":
            img_cls, vid_tkn = self.distillation_step(img, vid)
            return self._head(F.add(img_cls, vid_tkn))
        if mode == "sum_residual":
            # executed as written (:149-161): vid_cls is overwritten with normalize(img_cls), so the embedding is
            # 2 * normalize(img_cls); the video branch still runs (its output feeds no loss term)
            vid_cls = self.vid_step(vid)
            img_cls, _seq = self.img_step(img, vid_cls)
            n = F.l2_normalize(img_cls)
            return self._head(F.add(n, n))
        if mode == "post_sum":
            # the reference unpacks three values from the two-value distillation_step (:163-164, TypeError); intended:
            # image CLS + the video CLS *embedding* (before the joint encoder)
            vid_cls = self.vid_step(vid)
            img_cls, _vid_tkn = self.img_step(img, vid_cls)
            return self._head(F.add(img_cls, F.cast(vid_cls, img_cls.dtype)))
        if mode in ("frame", "pre_modal"):
            # pre_modal passes the bound method ``self.vid_step`` as distil_inject (:188); img_step only reads it in
            # "sum" mode, so the executed behaviour is the image-only path
            return self.img_step(img, None)
        if mode == "vid":
            return self._head(self.vid_step(vid))
        raise ValueError(f"unknown model mode {mode!r}")

    # ------------------------------------------------------------------ steps (:246-366)
    def _loss(self, batch):
        target, img, vid = batch
        target = target.float()
        mode = self.hparams.model
        if mode == "distil":
            s, t = self(img, vid)
            distil_loss = self.distil_criterion(s, t)
            base_loss = self.criterion(s, target)
            self.log("train/distilloss", distil_loss)
            self.log("train/bass_loss", base_loss)
            self.log("train/cossim", self.cos(s, t)[0])                        # :257
            return F.add(base_loss.reshape(1), distil_loss.reshape(1)).reshape(()), s
        if mode in ("sum", "sum_residual", "post_sum"):
            data = self(img, vid)
        elif mode == "pre_modal":
            data = self(img, None)
        elif mode == "frame":
            data = self(img, None)
        else:
            data = self(None, vid)
        return self.criterion(data, target), data

    def training_step(self, batch, batch_idx):
        loss, data = self._loss(batch)
        self.train_aprc(data, batch[0].int())                                  # :275-277
        self.log("train/loss", loss, on_step=False, on_epoch=True)
        self.log("train/aprc", self.train_aprc, on_step=False, on_epoch=True)
        return loss

    def _accumulate(self, data, target):
        """:331-334 / :363-366: the callbacks consume sigmoid probabilities and integer labels."""
        with torch.no_grad():
            self.running_logits.append(F.sigmoid(data.detach()))
        self.running_labels.append(target.int())

    def validation_step(self, batch, batch_idx):
        loss, data = self._loss(batch)
        self._accumulate(data, batch[0])
        self.val_aprc(data, batch[0].int())                                    # :335
        self.log("val/loss", loss, on_epoch=True)
        self.log("val/aprc", self.val_aprc, on_step=False, on_epoch=True)
        return loss

    def test_step(self, batch, batch_idx):
        loss, data = self._loss(batch)
        self._accumulate(data, batch[0])
        return loss
