"""Mirror of the reference's ``src/models/transformer.py`` ("PTN": one post-norm
transformer encoder per modality over pre-extracted 2048-d expert embeddings, CLS per
modality, sum fusion, LN + Linear head) on HIP kernels.  SURVEY section 8 row a19.

Kept: ``SimpleTransformer(**config)``, ``add_pos_cls``, ``ptn``, ``shared_step``,
``training_step(batch={"experts","label"})``, ``configure_optimizers`` (SGD,
transformer.py:59-60) and the state-dict names (``encoder_layers{0,1}.*`` templates,
``transformer_encoder{0,1}.layers.{i}.*``, ``norm``, ``cls``, ``mlp_head.{0,1}``,
``mlp_encoder.{0,1}``, ``position_encoder.pe``).  As in the reference only experts 0
and 1 are routed through an encoder (transformer.py:115-118); further experts contribute
their normalised CLS row unchanged.  ``forward`` / ``ptn_shared`` reference undefined
members in the reference (transformer.py:66-72) and are not reproduced.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as F
from .. import optim
from .. import ops
from ..lightning_compat import LightningModule
from .frame_transformer import Encoder, EncoderLayer, PositionalEncoding


class SimpleTransformer(LightningModule):
    def __init__(self, **kwargs):
        super(SimpleTransformer, self).__init__()
        self.save_hyperparameters()
        hp = self.hparams
        if hp.get("cls", 0):
            hp.seq_len = hp.seq_len + 1
        d = hp.input_dimension
        self.d = d
        self.compute_dtype = hp.get("compute_dtype", torch.bfloat16)
        self.criterion = F.bce_with_logits
        self.position_encoder = PositionalEncoding(d, hp.dropout, max_len=hp.seq_len)          # :35-36 (2048 there)
        self.encoder_layers0 = EncoderLayer(d, hp.nhead, hp.nhid, hp.dropout)                   # :39-42
        self.transformer_encoder0 = Encoder(d, hp.nhead, hp.nhid, hp.nlayers, hp.dropout)
        self.encoder_layers1 = EncoderLayer(d, hp.nhead, hp.nhid, hp.dropout)                   # :44-47
        self.transformer_encoder1 = Encoder(d, hp.nhead, hp.nhid, hp.nlayers, hp.dropout)
        self.norm = nn.LayerNorm(d)                                                             # :49
        self.running_labels = []
        self.running_logits = []
        self.cls = nn.Parameter(torch.rand(1, hp.batch_size, d))                                # :52-53
        self.mlp_head = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, hp.get("n_out", 15)))       # :54
        self.mlp_encoder = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, 1024))                   # :55-56

    def configure_optimizers(self):
        hp = self.hparams
        return optim.SGD(self.parameters(), lr=hp.learning_rate, momentum=hp.momentum,
                               weight_decay=hp.weight_decay)

    def add_pos_cls(self, data):
        """data [b, s, d] -> [s+1, b, d]: seq-first, prepend the learned CLS row, add the
        sinusoid table, LayerNorm (transformer.py:74-82; the b<->s rearranges around the
        row-wise LayerNorm are layout no-ops)."""
        seq = F.to_seq_first(data)
        seq = F.concat_rows(self.cls, seq)
        seq = self.position_encoder(seq)
        return F.layernorm(seq, self.norm.weight, self.norm.bias, self.norm.eps)

    def ptn(self, data):
        """data [BATCH, SEQ, EXPERTS, DIM] (transformer.py:106-133)."""
        B, S, E, D = data.shape
        data = data.contiguous()
        out = None
        for i in range(E):
            x = torch.empty((B, S, D), dtype=data.dtype, device=data.device)
            ops.copy2d(data[:, :, i], x, B * S, D, E * D, D)            # expert slice, 'b s e d -> e b s d'
            e = self.add_pos_cls(F.cast(x, self.compute_dtype))
            if i == 0:
                e = self.transformer_encoder0(e)
            elif i == 1:
                e = self.transformer_encoder1(e)
            cls = F.select_seq_first_row(e, 0)                          # e[:, 0, :] after 's b d -> b s d'
            out = cls if out is None else F.add(out, cls)               # sum over experts (:127-130)
        h = self.mlp_head
        return F.linear(F.layernorm(out, h[0].weight, h[0].bias, h[0].eps), h[1].weight, h[1].bias, out_f32=True)

    def shared_step(self, data):
        if self.hparams.model in ("ptn", "ptn_shared"):                 # both call ptn (:162-168)
            return self.ptn(data)
        raise NotImplementedError(f"model {self.hparams.model!r}")

    def training_step(self, batch, batch_idx):
        data = self.shared_step(batch["experts"])
        loss = self.criterion(data, batch["label"].float())
        self.log("train/loss", loss, on_step=True, on_epoch=True)
        return loss

    def validation_step(self, batch, batch_idx):
        data = self.shared_step(batch["experts"])
        loss = self.criterion(data, batch["label"].float())
        self.running_labels.append(batch["label"].int())
        with torch.no_grad():       # the reference appends sigmoid(data) AND data (:153-158: two entries per label batch,
            self.running_logits.append(F.sigmoid(data.detach()))   # which breaks its own callback); the probabilities are kept
        self.log("val/loss", loss, on_step=False, on_epoch=True)
        return loss
