"""torch.autograd.Function wrappers: the differentiable operators of the clip path.

Forward and backward of every Function are sequences of HIP kernel launches
through ``ops`` (the C ABI); torch contributes tensors, the autograd tape and
nothing else.  Parameters are fp32 masters; GEMM operands are cast to the
compute dtype (bf16 by default) inside the Functions and parameter gradients are
produced directly in fp32 (the weight-gradient GEMM writes fp32).

Block-level Functions (``attn_block``, ``mlp_block``) cover one residual branch
``fn(LN(x)) + x`` each (src/models/vit.py:71-74) so that the fusions that cross
operator boundaries live in one place: bias+GELU and bias+residual in the GEMM
epilogues, GELU' in the dgrad epilogue, and the sum of the two gradient paths
into ``x`` inside the LayerNorm backward kernel.
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import _lib as L
from . import ops

Tensor = torch.Tensor


def _wc(w: Optional[Tensor], dtype: torch.dtype) -> Optional[Tensor]:
    """Compute-dtype copy of an fp32 master weight (identity in fp32 mode).  When the
    parameter lives in a ``dp.FlatParameters`` buffer whose bf16 mirror is current,
    that mirror is used (one cast launch per step for all weights instead of one
    per layer)."""
    if w is None:
        return None
    c = getattr(w, "_dvt_compute", None)
    if c is not None and c.dtype == dtype and w._dvt_sink.owner.compute_valid:
        return c
    w = w.detach()
    if w.dtype == dtype:
        return w.contiguous()
    return ops.cast(w, dtype)


def _sink(p):
    """Gradient sink of a parameter managed by ``dp.FlatParameters`` (else None)."""
    return None if p is None else getattr(p, "_dvt_sink", None)


def _emit_wgrad(sink, dy2: Tensor, x2: Tensor):
    """dW = dy^T x, written into the sink (returns None) or returned as a new tensor."""
    if sink is None:
        return ops.linear_wgrad(dy2, x2)
    ops.linear_wgrad(dy2, x2, out=sink.buf.view(dy2.shape[1], x2.shape[1]), accumulate=not sink.fresh)
    sink.mark_written()
    return None


def _emit_wgrad_bias(sink_w, sink_b, dy2: Tensor, x2: Tensor, has_bias: bool, defer: bool = False):
    """(dW, db) from ONE pass over dy: the bias gradient rides on the weight-gradient GEMM.
    defer: -> (dW, db, pending, finish): the split-K reduce of the launch is left to the data-gradient GEMM that follows
    (``ops.linear_dgrad(..., carry=pending)`` performs it in the idle tail of its own grid); call ``finish()`` behind that
    call -- only then are dW / db complete and the sinks marked written (a bucket's all-reduce may start on that mark)."""
    N, K = dy2.shape[1], x2.shape[1]
    w_out = sink_w.buf.view(N, K) if sink_w is not None else None
    b_out, b_acc = None, False
    if has_bias:
        if sink_b is not None:
            b_out, b_acc = sink_b.buf.view(-1), not sink_b.fresh
        else:
            b_out = torch.empty((N,), dtype=torch.float32, device=dy2.device)
    r = ops.linear_wgrad(dy2, x2, out=w_out, accumulate=(not sink_w.fresh) if sink_w is not None else False,
                         bias_out=b_out, bias_accumulate=b_acc, defer_reduce=defer)
    dw, pending = r if defer else (r, None)

    def finish():
        if sink_w is not None:
            sink_w.mark_written()
        if has_bias and sink_b is not None:
            sink_b.mark_written()

    dw_ret = None if sink_w is not None else dw
    db_ret = None if (not has_bias or sink_b is not None) else b_out
    if defer:
        return dw_ret, db_ret, pending, finish
    finish()
    return dw_ret, db_ret


def _linear_bwd(sink_w, sink_b, dy2: Tensor, x2: Tensor, w: Tensor, has_bias: bool, *, epilogue: int = L.EPI_NONE,
                aux: Optional[Tensor] = None):
    """Backward of y = x2 w^T (+ bias): -> (dW, db, dx) with dW / db None when they went into their sinks.  One launch for
    launch-bound shapes, else weight gradient + data gradient with the split-K reduce carried (``ops.linear_backward``)."""
    N, K = dy2.shape[1], x2.shape[1]
    w_out = sink_w.buf.view(N, K) if sink_w is not None else None
    b_out, b_acc = None, False
    if has_bias:
        if sink_b is not None:
            b_out, b_acc = sink_b.buf.view(-1), not sink_b.fresh
        else:
            b_out = torch.empty((N,), dtype=torch.float32, device=dy2.device)
    dw, dx = ops.linear_backward(dy2, x2, w, out=w_out, accumulate=(not sink_w.fresh) if sink_w is not None else False,
                                 bias_out=b_out, bias_accumulate=b_acc, epilogue=epilogue, aux=aux)
    if sink_w is not None:
        sink_w.mark_written()
    if has_bias and sink_b is not None:
        sink_b.mark_written()
    return (None if sink_w is not None else dw), (None if (not has_bias or sink_b is not None) else b_out), dx


def _emit_colsum(sink, dy2: Tensor):
    if sink is None:
        return ops.colsum(dy2)
    ops.colsum(dy2, out=sink.buf.view(-1), accumulate=not sink.fresh)
    sink.mark_written()
    return None


# dgamma / dbeta of a LayerNorm whose gradients go into sinks: the reduce of the per-workgroup partial rows is deferred and
# performed for many layers by ONE launch (ops.layernorm_flush: at the end of every full-size attention block's backward, and
# in dp.FlatParameters.finish_backward) -- the sinks are marked written only then.
LN_DEFER = True
_ln_pending_sinks = set()


def ln_flush(end_of_step: bool = False):
    ops.layernorm_flush()
    _ln_pending_sinks.clear()
    if end_of_step:
        _lp_hand.clear()


def _ln_bwd(dy, x, g, mean, rstd, sg, sb, **kw):
    """LayerNorm backward with optional sinks for dgamma / dbeta -> (dx, dgamma, dbeta[, dx_lp])."""
    if sg is None or sb is None:
        return ops.layernorm_bwd(dy, x, g, mean, rstd, **kw)
    defer = None
    if LN_DEFER:
        if id(sg) in _ln_pending_sinks or id(sb) in _ln_pending_sinks:     # a second write to a sink still pending: in order
            ln_flush()
        # only a FIRST write is deferred: an accumulating one must be enqueued before its bucket's exchange can start, and
        # the bucket logic counts first writes only
        if sg.fresh and sb.fresh:
            _ln_pending_sinks.update((id(sg), id(sb)))

            def defer():
                sg.mark_written()
                sb.mark_written()
    # one flag per sink: gamma and beta may sit in different gradient buckets (dp.GradSink late-write redirect)
    r = ops.layernorm_bwd(dy, x, g, mean, rstd, dg=sg.buf.view(-1), db=sb.buf.view(-1),
                          accumulate=not sg.fresh, accumulate_beta=not sb.fresh, defer=defer, **kw)
    if defer is None:
        sg.mark_written()
        sb.mark_written()
    return (r[0], None, None) + tuple(r[3:])


def _emit_into(sink, t: Tensor):
    """Add (or store, when the sink is fresh) an fp32 temporary into a sink: the slow form for kernels that take one
    accumulate flag for two destinations whose sinks disagree (one of them redirected to its late buffer)."""
    ops.axpby_f32_(sink.buf.view(-1), t.reshape(-1).contiguous(), 1.0, 0.0 if sink.fresh else 1.0)
    sink.mark_written()


# ---- fp32 residual stream of the launch-bound zone (the 33-token temporal encoder, vit.py:122-128: 264 rows at B = 8).
# There single rows carry whole gradients and nothing averages the rounding of 16-bit storage, while fp32 storage of 264 rows
# costs no bandwidth: the blocks of that stack take and return an fp32 map (x fp32 -> LN -> 16-bit GEMM operands -> fp32
# residual add in the GEMM epilogue), and their gradients likewise.  A block's backward needs the incoming fp32 gradient as a
# 16-bit GEMM operand as well: the LayerNorm backward that produced it wrote that copy in the same pass and leaves it here,
# keyed by the gradient's address (a miss -- autograd handed a different tensor on -- costs one cast launch, not correctness).
# An entry HOLDS the fp32 tensor (so the caching allocator cannot hand its address to another gradient while the entry lives)
# and its version counter (an in-place change of the fp32 values after the copy was made is a miss); entries nobody took
# (the first temporal block's, whose consumer is not a block) are dropped at the start of the next forward (`lp_clear`) and in
# `ln_flush(end_of_step=True)`.
_lp_hand = {}


def lp_clear() -> None:
    _lp_hand.clear()


def _lp_put(t32: Tensor, lp: Tensor) -> None:
    _lp_hand[t32.data_ptr()] = (t32, t32._version, lp)


def _lp_take(t32: Tensor, T: torch.dtype) -> Tensor:
    e = _lp_hand.pop(t32.data_ptr(), None)
    if (e is not None and e[0].shape == t32.shape and e[0].stride() == t32.stride() and e[0].dtype == t32.dtype
            and t32._version == e[1] and e[2].dtype == T):
        return e[2]
    return ops.cast(t32, T)


def _f32(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    t = t.detach()
    return t.contiguous() if t.dtype == torch.float32 else ops.cast(t, torch.float32)


def _same_layout(a: Tensor, b: Tensor) -> bool:
    """Equal shapes and equal strides on every dimension of extent > 1."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


def _as(t: Tensor, dtype: torch.dtype) -> Tensor:
    return t if t.dtype == dtype else ops.cast(t, dtype)


# ---------------------------------------------------------------------------
# Linear (+ optional fp32 output), used for patch embedding and heads
# ---------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, out_f32):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        wc = _wc(w, x.dtype)
        M, K = x2.shape
        N = wc.shape[0]
        out_dtype = torch.float32 if out_f32 else x.dtype
        y = ops.gemm(x2, wc, M, N, K, a_kmajor=True, b_kmajor=True, lda=x2.stride(0), ldb=K,
                     out_dtype=out_dtype, bias=_f32(b))
        ctx.save_for_backward(x2, wc)
        ctx.has_bias = b is not None
        ctx.xshape = shp
        ctx.sinks = (_sink(w), _sink(b))
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, wc = ctx.saved_tensors
        dy2 = _as(dy.reshape(-1, dy.shape[-1]).contiguous(), x2.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(dy2, wc).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw, db = _emit_wgrad_bias(ctx.sinks[0], ctx.sinks[1], dy2, x2, ctx.has_bias and ctx.needs_input_grad[2])
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = _emit_colsum(ctx.sinks[1], dy2)
        return dx, dw, db, None


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None, out_f32: bool = False) -> Tensor:
    return _Linear.apply(x, w, b, out_f32)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.act_fwd(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(_as(dy, x.dtype), x, ctx.act), None


def gelu(x: Tensor) -> Tensor:
    return _Act.apply(x, ops.ACT_GELU)


def relu(x: Tensor) -> Tensor:
    return _Act.apply(x, ops.ACT_RELU)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a: Tensor, b: Tensor) -> Tensor:
    return _Add.apply(a, b)


class _Fork(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (a residual block's input feeds the convolution path and the
    shortcut, custom_resnet.py:38-54): the two incoming gradients are summed by ``dvt_add`` instead of autograd's own
    accumulate kernel, so that no ATen kernel runs inside the step."""

    @staticmethod
    def forward(ctx, x):
        return x.view(x.shape), x.view(x.shape)

    @staticmethod
    def backward(ctx, da, db):
        if da is None or db is None:
            return da if db is None else db
        if da.dtype != db.dtype:
            db = _as(db.contiguous(), da.dtype)
        return ops.add(da.contiguous(), db.contiguous())


def fork(x: Tensor):
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    return _Fork.apply(x)


class _Cast(torch.autograd.Function):
    """Differentiable dtype change of an activation (grad cast back)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return _as(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return _as(dy.contiguous(), ctx.src), None


def cast(x: Tensor, dtype: torch.dtype) -> Tensor:
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


# ---------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        xc = x.contiguous()
        g, bb = _f32(w), _f32(b)
        y, mean, rstd = ops.layernorm_fwd(xc, g, bb, eps)
        ctx.save_for_backward(xc, g, mean, rstd)
        ctx.sinks = (_sink(w), _sink(b))
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, g, mean, rstd = ctx.saved_tensors
        dx, dg, db = _ln_bwd(_as(dy.contiguous(), xc.dtype), xc, g, mean, rstd, *ctx.sinks)
        return dx, dg, db, None


def layernorm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    return _LayerNorm.apply(x, w, b, eps)


# ---------------------------------------------------------------------------
# Patch embedding: patchify gather + Linear  (vit.py:89-92,110)
# ---------------------------------------------------------------------------
class _PatchEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip, w, b, patch, dtype):
        patches = ops.patchify(clip, patch, dtype)          # [frames*n, P*P*C]
        wc = _wc(w, dtype)
        emb = ops.linear_fwd(patches, wc, _f32(b))
        ctx.save_for_backward(patches, wc)
        ctx.clip_shape, ctx.clip_dtype, ctx.patch = tuple(clip.shape), clip.dtype, patch
        ctx.sinks = (_sink(w), _sink(b))
        return emb

    @staticmethod
    def backward(ctx, demb):
        patches, wc = ctx.saved_tensors
        demb = demb.contiguous()
        dclip = None
        if ctx.needs_input_grad[0]:
            dpatch = ops.linear_dgrad(demb, wc)
            dclip = ops.patchify_bwd(dpatch, ctx.clip_shape, ctx.patch, ctx.clip_dtype)
        dw, db = _emit_wgrad_bias(ctx.sinks[0], ctx.sinks[1], demb, patches, True)
        return dclip, dw, db, None, None


def patch_embed(clip: Tensor, w: Tensor, b: Tensor, patch: int, dtype: torch.dtype) -> Tensor:
    """clip [..., C, H, W] -> [frames * n, d] in ``dtype``."""
    return _PatchEmbed.apply(clip, w, b, patch, dtype)


# ---------------------------------------------------------------------------
# Token assembly (CLS concat + learned positional add), vit.py:113-115
# ---------------------------------------------------------------------------
class _Tokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, cls, pos, S, T, n):
        cls32 = _f32(cls).reshape(-1)
        pos32 = _f32(pos)
        pos32 = pos32.reshape(T, pos32.shape[-2], pos32.shape[-1])
        ctx.T, ctx.pos_rows = T, pos32.shape[1]
        ctx.cls_shape, ctx.pos_shape = tuple(cls.shape), tuple(pos.shape)
        ctx.sinks = (_sink(cls), _sink(pos))
        return ops.tokens_assemble_fwd(emb.contiguous(), cls32, pos32, S, T, n)

    @staticmethod
    def backward(ctx, dout):
        sc, sp = ctx.sinks
        if sc is not None and sp is not None and sc.fresh != sp.fresh:
            demb, dcls, dpos = ops.tokens_assemble_bwd(dout, ctx.T, ctx.pos_rows)
            _emit_into(sc, dcls)
            _emit_into(sp, dpos)
            return demb, None, None, None, None, None
        if sc is not None and sp is not None:
            demb, _, _ = ops.tokens_assemble_bwd(dout, ctx.T, ctx.pos_rows, dcls=sc.buf.view(-1),
                                                 dpos=sp.buf.view(ctx.T, ctx.pos_rows, -1),
                                                 accumulate=not sc.fresh)
            sc.mark_written()
            sp.mark_written()
            return demb, None, None, None, None, None
        demb, dcls, dpos = ops.tokens_assemble_bwd(dout, ctx.T, ctx.pos_rows)
        return demb, dcls.view(ctx.cls_shape), dpos.view(ctx.pos_shape), None, None, None


def tokens_assemble(emb: Tensor, cls: Tensor, pos: Tensor, S: int, T: int, n: int) -> Tensor:
    """emb [S*n, d], cls [1,1,d], pos [1,T,rows,d] -> [S, n+1, d]."""
    return _Tokens.apply(emb, cls, pos, S, T, n)


# ---------------------------------------------------------------------------
# Final space LayerNorm on the CLS rows + temporal CLS concat (vit.py:75,120-123)
# ---------------------------------------------------------------------------
class _ClsNormConcat(torch.autograd.Function):
    """x [B*T, N, d] -> LN(x)[:, 0] as [B, T, d] -> cat(token, .) -> [B, T+1, d].

    LayerNorm is row-wise, so normalising only the rows that are read
    (``x[:, 0]``, vit.py:120) gives the same values and gradients as the
    reference's LayerNorm over all N rows followed by the slice.
    """

    @staticmethod
    def forward(ctx, x, w, b, tok, B, T, eps, out_dtype=None):
        # out_dtype float32 (x 16-bit): the sequence enters the fp32 residual stream of the temporal stack
        S, N, d = x.shape
        xc = x.contiguous()
        g, bb = _f32(w), _f32(b)
        rows = (S, 1, N * d, 0)
        cls_rows, mean, rstd = ops.layernorm_fwd(xc, g, bb, eps, rows=rows, out_dtype=out_dtype)      # [S, d]
        tok32 = _f32(tok).reshape(-1) if tok is not None else None
        seq = ops.rows_gather_fwd(cls_rows, d, tok32, B, T, d)
        ctx.save_for_backward(xc, g, mean, rstd)
        ctx.dims = (B, T, S, N, d)
        ctx.seq_dtype = seq.dtype
        ctx.tok_shape = tuple(tok.shape) if tok is not None else None
        ctx.sinks = (_sink(w), _sink(b), _sink(tok))
        return seq

    @staticmethod
    def backward(ctx, dseq):
        xc, g, mean, rstd = ctx.saved_tensors
        B, T, S, N, d = ctx.dims
        gdt = ctx.seq_dtype                               # the sequence's type: xc's, or fp32 (then dy fp32 -> dx 16-bit)
        dseq = _as(dseq.contiguous(), gdt)
        drows = torch.empty((S, d), dtype=gdt, device=xc.device)
        sg, sb, st = ctx.sinks
        if st is not None:
            ops.rows_gather_bwd(dseq, drows, d, True, B, T, d, dtok=st.buf.view(-1), accumulate=not st.fresh)
            st.mark_written()
            dtok = None
        else:
            dtok = ops.rows_gather_bwd(dseq, drows, d, ctx.tok_shape is not None, B, T, d)
        # rows other than CLS receive no gradient (N == 1, the folded last layer: there are none, nothing to clear)
        dx = ops.zeros(xc.shape, xc.dtype, xc.device) if N > 1 else torch.empty_like(xc)
        _, dg, db = _ln_bwd(drows, xc, g, mean, rstd, sg, sb, rows=(S, 1, N * d, 0), dy_rows=(d, 0), dx=dx)
        if dtok is not None:
            dtok = dtok.view(ctx.tok_shape)
        return dx, dg, db, dtok, None, None, None, None


def cls_norm_concat(x: Tensor, w: Tensor, b: Tensor, tok: Optional[Tensor], B: int, T: int,
                    eps: float = 1e-5, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    return _ClsNormConcat.apply(x, w, b, tok, B, T, eps, out_dtype)


class _RowsSelect(torch.autograd.Function):
    """x [B, L, d] -> x[:, 0] (pool == 'cls', vit.py:126) without a torch kernel."""

    @staticmethod
    def forward(ctx, x):
        B, Ln, d = x.shape
        xc = x.contiguous()
        ctx.shape = (B, Ln, d)
        return ops.rows_gather_fwd(xc, Ln * d, None, B, 1, d).view(B, d)

    @staticmethod
    def backward(ctx, dy):
        B, Ln, d = ctx.shape
        dx = ops.zeros((B, Ln, d), dy.dtype, dy.device)
        ops.rows_gather_bwd(dy.contiguous().view(B, 1, d), dx, Ln * d, False, B, 1, d)
        return dx


def select_first_row(x: Tensor) -> Tensor:
    return _RowsSelect.apply(x)


# ---------------------------------------------------------------------------
# Attention core on separate q/k/v (cross-modal form; MHA)
# ---------------------------------------------------------------------------
class _AttentionCore(torch.autograd.Function):
    """q [B,H,Lq,dh], k/v [B,H,Lk,dh] (arbitrary strides, dh contiguous) ->
    o [B,H,Lq,dh] laid out as [B, Lq, H, dh] in memory ('b n (h d)', vit.py:56)."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        B, H, Lq, dh = q.shape
        o_mem = torch.empty((B, Lq, H, dh), dtype=q.dtype, device=q.device)
        o = o_mem.permute(0, 2, 1, 3)
        lse = ops.attention_fwd(q, k, v, o, scale)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        B, H, Lq, dh = q.shape
        Lk = k.shape[2]
        if _same_layout(do, o) and do.dtype == q.dtype:
            do_v = do.as_strided(o.shape, o.stride())
        else:
            # re-layout into o's memory order ([B, Lq, H, dh]) before the kernel
            do_c = _as(do.permute(0, 2, 1, 3).contiguous(), q.dtype)
            do_v = do_c.permute(0, 2, 1, 3).as_strided(o.shape, o.stride())
        dq = torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=q.device)
        dk = torch.empty_strided(k.shape, k.stride(), dtype=k.dtype, device=k.device)
        dv = torch.empty_strided(v.shape, v.stride(), dtype=v.dtype, device=v.device)
        ops.attention_bwd(q, k, v, o, lse, do_v, dq, dk, dv, ctx.scale)
        return dq, dk, dv, None


def attention_core(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    return _AttentionCore.apply(q, k, v, scale)


# ---------------------------------------------------------------------------
# Residual branches of the pre-norm Transformer (vit.py:30-75)
# ---------------------------------------------------------------------------
def _split_qkv(qkv: Tensor, S: int, N: int, heads: int, dh: int, seq_first: bool):
    """Packed [rows, 3*h*dh] -> three [S(batch), H, N(seq), dh] strided views (no copies).
    Rows are (s n) for batch-first input and (n s) for seq-first input."""
    if seq_first:
        q5 = qkv.view(N, S, 3, heads, dh)
        return tuple(q5[:, :, i].permute(1, 2, 0, 3) for i in range(3))
    q5 = qkv.view(S, N, 3, heads, dh)
    return tuple(q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))


def _heads_view(o_mem: Tensor, seq_first: bool) -> Tensor:
    """[S,N,H,dh] (or [N,S,H,dh] seq-first) memory -> [S, H, N, dh] view."""
    return o_mem.permute(1, 2, 0, 3) if seq_first else o_mem.permute(0, 2, 1, 3)


class _AttnBlock(torch.autograd.Function):
    """y = [x +] to_out(attention(to_qkv([LN](x))))   -- ``PreNorm(Attention)`` + residual.

    to_qkv has no bias (vit.py:39); q,k,v are the three chunks of the last dim and
    heads are the outer factor of each chunk (vit.py:48-49); to_out is skipped when
    ``w_out is None`` (heads == 1 and dim_head == dim, vit.py:34,41-44).
    """

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w_qkv, w_out, b_out, heads, prenorm, residual, eps, b_qkv=None,
                seq_first=False, attn_dropout=0.0, cdt=None):
        # seq_first: x is [L, B, E] (torch nn.MultiheadAttention layout, frame_transformer.py:204-207)
        # cdt: element type of the GEMM operands when x is an fp32 stream (see _lp_hand); None: x's own
        shp = x.shape
        d = shp[-1]
        if seq_first:
            N, S = shp[0], shp[1]          # N = sequence length, S = batch
        else:
            S, N = (shp[0], shp[1]) if x.dim() == 3 else (1, shp[0])
        x2 = x.reshape(-1, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        T = x.dtype if cdt is None else cdt
        ctx.mixed = mixed = T != x.dtype
        if mixed and not (x.dtype == torch.float32 and prenorm and residual and w_out is not None and attn_dropout == 0.0):
            raise ValueError("the fp32 stream form is the pre-norm residual block with an output projection")
        if prenorm:
            g, bb = _f32(ln_w), _f32(ln_b)
            xn, mean, rstd = ops.layernorm_fwd(x2, g, bb, eps, out_dtype=T)
        else:
            g = mean = rstd = None
            xn = x2
        wq = _wc(w_qkv, T)
        inner = wq.shape[0] // 3
        dh = inner // heads
        qkv = ops.linear_fwd(xn, wq, _f32(b_qkv))                      # [M, 3*inner]
        q, k, v = _split_qkv(qkv, S, N, heads, dh, seq_first)          # [S,H,N,dh] views
        o_mem = torch.empty((N, S, heads, dh) if seq_first else (S, N, heads, dh), dtype=T, device=x.device)
        drop = None
        if attn_dropout > 0.0:                        # nn.MultiheadAttention(dropout=p): Philox mask on the probabilities
            drop = (attn_dropout, _rng.tensor(x.device), _rng.take(S * heads * N * N))
        lse = ops.attention_fwd(q, k, v, _heads_view(o_mem, seq_first), dh ** -0.5, drop)
        ctx.drop = drop
        o2 = o_mem.view(M, inner)
        if w_out is not None:
            wo = _wc(w_out, T)
            if residual:
                y = ops.linear_fwd(o2, wo, _f32(b_out), epilogue=L.EPI_RESIDUAL, residual=x2,
                                   out_dtype=torch.float32 if mixed else None)
            else:
                y = ops.linear_fwd(o2, wo, _f32(b_out))
        else:
            wo = None
            y = ops.add(o2, x2) if residual else o2
        ctx.save_for_backward(x2, g, mean, rstd, xn if prenorm else None, wq, wo, qkv, o_mem, lse)
        ctx.cfg = (S, N, heads, dh, inner, prenorm, residual, w_out is not None, b_out is not None,
                   b_qkv is not None, seq_first)
        ctx.xshape = shp
        ctx.sinks = tuple(_sink(t) for t in (ln_w, ln_b, w_qkv, w_out, b_out, b_qkv))
        return y.view(*shp[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, g, mean, rstd, xn, wq, wo, qkv, o_mem, lse = ctx.saved_tensors
        S, N, heads, dh, inner, prenorm, residual, has_out, has_bias, has_qkv_bias, seq_first = ctx.cfg
        M = x2.shape[0]
        T = qkv.dtype
        if xn is None:
            xn = x2
        if ctx.mixed:                                    # fp32 gradient stream: its 16-bit copy is the GEMM operand
            dy32 = dy.reshape(M, -1).contiguous()
            dy2 = _lp_take(dy32, T)
        else:
            dy2 = _as(dy.reshape(M, -1).contiguous(), T)
        dwo = dbo = None
        if has_out:
            dwo, dbo, do2 = _linear_bwd(ctx.sinks[3], ctx.sinks[4], dy2, o_mem.view(M, inner), wo, has_bias)   # do2 [M, inner]
        else:
            do2 = dy2
        q, k, v = _split_qkv(qkv, S, N, heads, dh, seq_first)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = _split_qkv(dqkv, S, N, heads, dh, seq_first)
        ops.attention_bwd(q, k, v, _heads_view(o_mem, seq_first), lse,
                          _heads_view(do2.view(o_mem.shape), seq_first), dq, dk, dv, dh ** -0.5, ctx.drop)
        dwq, dbq, dxn = _linear_bwd(ctx.sinks[2], ctx.sinks[5], dqkv, xn, wq, has_qkv_bias)                    # dxn [M, d]
        dg = db = None
        if ctx.mixed:
            dx, dg, db, dx_lp = _ln_bwd(dxn, x2, g, mean, rstd, ctx.sinks[0], ctx.sinks[1], dx_add=dy32,
                                        dx_dtype=torch.float32, dx_lp=T)
            _lp_put(dx, dx_lp)
        elif prenorm:
            dx, dg, db = _ln_bwd(dxn, x2, g, mean, rstd, ctx.sinks[0], ctx.sinks[1],
                                 dx_add=dy2 if residual else None)
        else:
            dx = ops.add(dxn, dy2) if residual else dxn
        if M >= 4096:                                    # a full-size layer: its two LayerNorms' reduces (and whatever the
            ln_flush()                                   # launch-bound zone left pending) in one launch
        return dx.view(ctx.xshape), dg, db, dwq, dwo, dbo, None, None, None, None, dbq, None, None, None


def attn_block(x, ln_w, ln_b, w_qkv, w_out, b_out, heads, *, prenorm=True, residual=True, eps=1e-5,
               b_qkv=None, seq_first=False, attn_dropout=0.0, cdt=None):
    """attn_dropout > 0: dropout on the attention probabilities (training mode of nn.MultiheadAttention).
    cdt: GEMM operand type when x is an fp32 residual stream (the launch-bound zone)."""
    return _AttnBlock.apply(x, ln_w, ln_b, w_qkv, w_out, b_out, heads, prenorm, residual, eps, b_qkv, seq_first,
                            float(attn_dropout), cdt)


class _AttnBlockCls(torch.autograd.Function):
    """Row 0 of ``x + to_out(attention(to_qkv(LN(x))))`` for every sequence: x [S, N, d] -> [S, d].

    The reference keeps only ``x[:, 0]`` of the space transformer's output (vit.py:119-120) and only
    ``x[:, 0]`` of the temporal one under ``pool == 'cls'`` (:126), so in the LAST layer of a stack
    everything behind the keys and values is needed for the first row alone: LayerNorm and the K / V
    projection run on all rows, the Q projection, the attention (one query per head), the output
    projection and the residual on S rows.  Values and gradients equal the dense block followed by the
    slice: the rows that are dropped contribute exact zeros to every gradient.  Backward: dK / dV come
    from the one query; the gradient into LN(x) is ``dkv W_kv`` on all rows plus ``dq W_q`` on row 0,
    summed inside the LayerNorm backward kernel together with the residual path of row 0.

    Folded form (16-bit dtypes, ``S * N >= CLS_FOLD_MIN_ROWS``, shapes ``ops.attn_cls_supported`` accepts: the space
    stack): with one query per (sequence, head) the K / V projections commute with the attention sums --
    ``s_jh = r_h . LN(x_j)`` with ``r_h = scale Wk_h^T q_h`` and ``o_h = Wv_h m_h`` with ``m_h = sum_j p_jh LN(x_j)`` -- so
    LN(x), K and V of the rows 1 .. N-1 never exist: one pass over the raw rows forward (``ops.attn_cls_fwd``), one more
    backward (``ops.attn_cls_bwd``, which also is those rows' LayerNorm backward), head-wise products over the S first
    rows on either side, and a LayerNorm backward call on the S first rows for the query / residual paths of row 0.
    Same values up to summation order (csrc/attention_cls.hip).
    """

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w_qkv, w_out, b_out, heads, eps, cdt=None):
        S, N, d = x.shape
        T = x.dtype if cdt is None else cdt
        ctx.mixed = mixed = T != x.dtype                 # fp32 residual stream (the temporal stack): unfolded form only
        x2 = x.reshape(S * N, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        g, bb = _f32(ln_w), _f32(ln_b)
        wqkv = _wc(w_qkv, T)
        inner = wqkv.shape[0] // 3
        dh = inner // heads
        wo = _wc(w_out, T)
        ctx.cfg = (S, N, d, heads, dh, inner, b_out is not None, eps)
        ctx.sinks = tuple(_sink(t) for t in (ln_w, ln_b, w_qkv, w_out, b_out))
        x3 = x2.view(S, N, d)
        ctx.folded = (not mixed) and cls_fold_taken(x3, heads, dh)
        if ctx.folded:
            # K / V projections folded into the query (csrc/attention_cls.hip): LN(x), K and V of the rows 1 .. N-1 never exist
            xn0, mean0, rstd0 = ops.layernorm_fwd(x2, g, bb, eps, rows=(S, 1, N * d, 0))          # [S, d]: row 0 only
            q = ops.linear_fwd(xn0, wqkv[:inner])                                                  # [S, inner]
            R = ops.heads_expand(q, wqkv[inner:2 * inner], heads, dh ** -0.5)                      # r_h = scale Wk_h^T q_h
            A, lse, P, mean, rstd = ops.attn_cls_fwd(x3, g, bb, eps, R)                            # one pass over x
            o = ops.heads_contract(A, wqkv[2 * inner:], 1.0, g, bb)                                # o_h = Wv_h (gamma A_h + beta)
            y = ops.linear_fwd(o, wo, _f32(b_out), epilogue=L.EPI_RESIDUAL, residual=x3[:, 0])
            ctx.save_for_backward(x2, g, bb, mean0, rstd0, xn0, wqkv, wo, q, o, R, A, lse, P, mean, rstd)
            return y
        xn, mean, rstd = ops.layernorm_fwd(x2, g, bb, eps, out_dtype=T)
        w_q, w_kv = wqkv[:inner], wqkv[inner:]                          # row slices of the packed weight, no copies
        kv = ops.linear_fwd(xn, w_kv)                                    # [S*N, 2*inner]
        xn0 = xn.view(S, N, d)[:, 0]                                     # [S, d], row stride N*d
        q = ops.linear_fwd(xn0, w_q)                                     # [S, inner]
        kv5 = kv.view(S, N, 2, heads, dh)
        k4, v4 = kv5[:, :, 0].permute(0, 2, 1, 3), kv5[:, :, 1].permute(0, 2, 1, 3)
        o = torch.empty((S, inner), dtype=T, device=x.device)
        lse = ops.attention_fwd(q.view(S, 1, heads, dh).permute(0, 2, 1, 3), k4, v4,
                                o.view(S, 1, heads, dh).permute(0, 2, 1, 3), dh ** -0.5)
        y = ops.linear_fwd(o, wo, _f32(b_out), epilogue=L.EPI_RESIDUAL, residual=x3[:, 0],
                           out_dtype=torch.float32 if mixed else None)
        ctx.save_for_backward(x2, g, mean, rstd, xn, wqkv, wo, q, kv, o, lse)
        return y

    @staticmethod
    def _backward_folded(ctx, dy):
        x2, g, bb, mean0, rstd0, xn0, wqkv, wo, q, o, R, A, lse, P, mean, rstd = ctx.saved_tensors
        S, N, d, heads, dh, inner, has_bias, eps = ctx.cfg
        T = x2.dtype
        s_g, s_b, s_qkv, s_o, s_bo = ctx.sinks
        scale = dh ** -0.5
        dy2 = _as(dy.reshape(S, d).contiguous(), T)
        dwo, dbo, do = _linear_bwd(s_o, s_bo, dy2, o, wo, has_bias)      # do [S, inner]
        if s_qkv is not None:                                            # three row ranges of one packed gradient
            buf, acc = s_qkv.buf.view(3 * inner, d), not s_qkv.fresh
            dwqkv = None
        else:
            buf, acc = torch.empty((3 * inner, d), dtype=torch.float32, device=x2.device), False
            dwqkv = buf
        w_q, w_k, w_v = wqkv[:inner], wqkv[inner:2 * inner], wqkv[2 * inner:]
        # dm_h = Wv_h^T do_h and dWv_h = do_h (x) m_h: independent, one launch
        dM = ops.heads_expand_outer(do, w_v, heads, A, buf[2 * inner:], gamma=g, beta=bb, accumulate=acc)
        if s_g is not None and s_b is not None:
            dx2, G, _, _ = ops.attn_cls_bwd(x2.view(S, N, d), g, bb, eps, R, A, lse, P, mean, rstd, dM, dg=s_g.buf.view(-1),
                                            db=s_b.buf.view(-1), accumulate=not s_g.fresh, accumulate_beta=not s_b.fresh)
            dg, db = s_g.buf.view(-1), s_b.buf.view(-1)
        else:
            dx2, G, dg, db = ops.attn_cls_bwd(x2.view(S, N, d), g, bb, eps, R, A, lse, P, mean, rstd, dM)
        # dq_h = scale Wk_h (gamma G_h) and dWk_h = scale q_h (x) (gamma G_h): independent, one launch
        dq = ops.heads_contract_outer(G, g, w_k, q, buf[inner:2 * inner], alpha_out=scale, alpha_dw=scale, accumulate=acc)
        _, dxn0 = ops.linear_backward(dq, xn0, w_q, out=buf[:inner], accumulate=acc)   # dWq; dxn0 [S, d]: the query path into LN(x)[:, 0]
        if s_qkv is not None:
            s_qkv.mark_written()
        # row 0 of every sequence: the LayerNorm backward of the query path, plus the residual path, on top of its K / V part
        dxf = dx2.view(S * N, d)
        ops.layernorm_bwd(dxn0, x2, g, mean0, rstd0, rows=(S, 1, N * d, 0), dx=dxf, dx_add=dxf, dx_first=dy2,
                          dg=dg, db=db, accumulate=True)
        if s_g is not None and s_b is not None:
            s_g.mark_written()
            s_b.mark_written()
            dg = db = None
        return dx2.view(S, N, d), dg, db, dwqkv, dwo, dbo, None, None, None

    @staticmethod
    def backward(ctx, dy):
        if ctx.folded:
            return _AttnBlockCls._backward_folded(ctx, dy)
        x2, g, mean, rstd, xn, wqkv, wo, q, kv, o, lse = ctx.saved_tensors
        S, N, d, heads, dh, inner, has_bias, _ = ctx.cfg
        T = kv.dtype
        s_g, s_b, s_qkv, s_o, s_bo = ctx.sinks
        dy_s = dy.reshape(S, d).contiguous()             # the stream's own type (fp32 in the mixed form): row 0's residual path
        dy2 = _lp_take(dy_s, T) if ctx.mixed else _as(dy_s, T)
        dwo, dbo, do = _linear_bwd(s_o, s_bo, dy2, o, wo, has_bias)      # do [S, inner]
        kv5 = kv.view(S, N, 2, heads, dh)
        k4, v4 = kv5[:, :, 0].permute(0, 2, 1, 3), kv5[:, :, 1].permute(0, 2, 1, 3)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        dkv5 = dkv.view(S, N, 2, heads, dh)
        one = lambda t: t.view(S, 1, heads, dh).permute(0, 2, 1, 3)
        ops.attention_bwd(one(q), k4, v4, one(o), lse, one(do), one(dq),
                          dkv5[:, :, 0].permute(0, 2, 1, 3), dkv5[:, :, 1].permute(0, 2, 1, 3), dh ** -0.5)
        xn0 = xn.view(S, N, d)[:, 0]
        w_q, w_kv = wqkv[:inner], wqkv[inner:]
        if s_qkv is not None:                                            # the two row ranges of one packed gradient
            buf, acc = s_qkv.buf.view(3 * inner, d), not s_qkv.fresh
            dwqkv = None
        else:
            buf, acc = torch.empty((3 * inner, d), dtype=torch.float32, device=x2.device), False
            dwqkv = buf
        ops.linear_wgrad(dq, xn0, out=buf[:inner], accumulate=acc)
        _, pend = ops.linear_wgrad(dkv, xn, out=buf[inner:], accumulate=acc, defer_reduce=True)
        dxn = ops.linear_dgrad(dkv, w_kv, carry=pend)                    # [S*N, d]: the K / V path, all rows (+ that reduce)
        if s_qkv is not None:
            s_qkv.mark_written()
        dxn0 = ops.linear_dgrad(dq, w_q)                                 # [S, d]: the Q path, row 0
        if ctx.mixed:
            dx, dg, db, dx_lp = _ln_bwd(dxn, x2, g, mean, rstd, s_g, s_b, rows=(S, N, N * d, d), dy_first=dxn0,
                                        dx_first=dy_s, dx_dtype=torch.float32, dx_lp=T)
            _lp_put(dx.view(S * N, d), dx_lp.view(S * N, d))
        else:
            dx, dg, db = _ln_bwd(dxn, x2, g, mean, rstd, s_g, s_b, rows=(S, N, N * d, d),
                                 dy_first=dxn0, dx_first=dy2)
        return dx.view(S, N, d), dg, db, dwqkv, dwo, dbo, None, None, None


# Sequences x rows from which the last layer's single-query attention runs with the K / V projections folded into the query
# (csrc/attention_cls.hip); below it (the 33-token temporal stack) the unfolded form has fewer launches.
CLS_FOLD_MIN_ROWS = 4096


def cls_fold_taken(x3: Tensor, heads: int, dh: int) -> bool:
    """The predicate ``_AttnBlockCls`` uses to take the folded form (bench.py prices ``executed_tflops`` with it)."""
    S, N, _ = x3.shape
    return S * N >= CLS_FOLD_MIN_ROWS and ops.attn_cls_supported(x3, heads, dh)


def attn_block_cls(x, ln_w, ln_b, w_qkv, w_out, b_out, heads, *, eps=1e-5, cdt=None):
    """``attn_block(x, ...)[:, 0]`` for x [S, N, d] without the rows that are never read."""
    return _AttnBlockCls.apply(x, ln_w, ln_b, w_qkv, w_out, b_out, heads, eps, cdt)


class _CrossAttnBlock(torch.autograd.Function):
    """y = x + to_out(attention(q = to_q(LN_q(x)), k,v = to_kv(LN_kv(c))))  -- the cross-modal attention
    block (BASELINE configs[3]: video tokens attend to audio tokens; Lq != Lk).  x [B, Lq, d], c [B, Lk, dc];
    to_q / to_kv have no bias (same convention as vit.py:39); kv is kept packed [B*Lk, 2*h*dh] and read by the
    attention kernel through strides, like the self-attention form."""

    @staticmethod
    def forward(ctx, x, c, lnq_w, lnq_b, lnk_w, lnk_b, w_q, w_kv, w_out, b_out, heads, eps):
        B, Lq, d = x.shape
        Lk = c.shape[1]
        T = x.dtype
        x2 = x.reshape(B * Lq, d).contiguous()
        c2 = _as(c.reshape(B * Lk, c.shape[-1]).contiguous(), T)
        gq, bq, gk, bk = _f32(lnq_w), _f32(lnq_b), _f32(lnk_w), _f32(lnk_b)
        xn, mq, rq = ops.layernorm_fwd(x2, gq, bq, eps)
        cn, mk, rk = ops.layernorm_fwd(c2, gk, bk, eps)
        wq, wkv, wo = _wc(w_q, T), _wc(w_kv, T), _wc(w_out, T)
        inner = wq.shape[0]
        dh = inner // heads
        q = ops.linear_fwd(xn, wq)                                       # [B*Lq, inner]
        kv = ops.linear_fwd(cn, wkv)                                     # [B*Lk, 2*inner]
        q4 = q.view(B, Lq, heads, dh).permute(0, 2, 1, 3)
        kv5 = kv.view(B, Lk, 2, heads, dh)
        k4, v4 = kv5[:, :, 0].permute(0, 2, 1, 3), kv5[:, :, 1].permute(0, 2, 1, 3)
        o_mem = torch.empty((B, Lq, heads, dh), dtype=T, device=x.device)
        lse = ops.attention_fwd(q4, k4, v4, o_mem.permute(0, 2, 1, 3), dh ** -0.5)
        y = ops.linear_fwd(o_mem.view(B * Lq, inner), wo, _f32(b_out), epilogue=L.EPI_RESIDUAL, residual=x2)
        ctx.save_for_backward(x2, c2, gq, gk, mq, rq, mk, rk, xn, cn, wq, wkv, wo, q, kv, o_mem, lse)
        ctx.cfg = (B, Lq, Lk, heads, dh, inner, tuple(x.shape), tuple(c.shape), c.dtype)
        ctx.sinks = tuple(_sink(t) for t in (lnq_w, lnq_b, lnk_w, lnk_b, w_q, w_kv, w_out, b_out))
        return y.view(B, Lq, d)

    @staticmethod
    def backward(ctx, dy):
        x2, c2, gq, gk, mq, rq, mk, rk, xn, cn, wq, wkv, wo, q, kv, o_mem, lse = ctx.saved_tensors
        B, Lq, Lk, heads, dh, inner, xshape, cshape, cdt = ctx.cfg
        s = ctx.sinks
        T = x2.dtype
        dy2 = _as(dy.reshape(B * Lq, -1).contiguous(), T)
        do2 = ops.linear_dgrad(dy2, wo)
        dwo, dbo = _emit_wgrad_bias(s[6], s[7], dy2, o_mem.view(B * Lq, inner), True)
        q4 = q.view(B, Lq, heads, dh).permute(0, 2, 1, 3)
        kv5 = kv.view(B, Lk, 2, heads, dh)
        k4, v4 = kv5[:, :, 0].permute(0, 2, 1, 3), kv5[:, :, 1].permute(0, 2, 1, 3)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        dkv5 = dkv.view(B, Lk, 2, heads, dh)
        ops.attention_bwd(q4, k4, v4, o_mem.permute(0, 2, 1, 3), lse, do2.view(o_mem.shape).permute(0, 2, 1, 3),
                          dq.view(B, Lq, heads, dh).permute(0, 2, 1, 3), dkv5[:, :, 0].permute(0, 2, 1, 3),
                          dkv5[:, :, 1].permute(0, 2, 1, 3), dh ** -0.5)
        dwq = _emit_wgrad(s[4], dq, xn)
        dwkv = _emit_wgrad(s[5], dkv, cn)
        dx, dgq, dbq = _ln_bwd(ops.linear_dgrad(dq, wq), x2, gq, mq, rq, s[0], s[1], dx_add=dy2)
        dc = dgk = dbk = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[4]:
            dc, dgk, dbk = _ln_bwd(ops.linear_dgrad(dkv, wkv), c2, gk, mk, rk, s[2], s[3])
            dc = _as(dc, cdt).view(cshape)
        return dx.view(xshape), dc, dgq, dbq, dgk, dbk, dwq, dwkv, dwo, dbo, None, None


def cross_attn_block(x, c, lnq_w, lnq_b, lnk_w, lnk_b, w_q, w_kv, w_out, b_out, heads, eps=1e-5):
    return _CrossAttnBlock.apply(x, c, lnq_w, lnq_b, lnk_w, lnk_b, w_q, w_kv, w_out, b_out, heads, eps)


class _MlpBlock(torch.autograd.Function):
    """y = [x +] W2 act(W1 [LN](x) + b1) + b2   -- ``PreNorm(FeedForward)`` + residual
    (vit.py:17-28,73-74).  act: 'gelu' (exact erf) or 'relu'."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, act, prenorm, residual, eps, cdt=None):
        shp = x.shape
        d = shp[-1]
        x2 = x.reshape(-1, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = x.dtype if cdt is None else cdt
        ctx.mixed = mixed = T != x.dtype                 # fp32 residual stream, 16-bit GEMM operands (see _lp_hand)
        if mixed and not (x.dtype == torch.float32 and prenorm and residual):
            raise ValueError("the fp32 stream form is the pre-norm residual block")
        if prenorm:
            g, bb = _f32(ln_w), _f32(ln_b)
            xn, mean, rstd = ops.layernorm_fwd(x2, g, bb, eps, out_dtype=T)
        else:
            g = mean = rstd = None
            xn = x2
        w1c, w2c = _wc(w1, T), _wc(w2, T)
        M = x2.shape[0]
        if act == "gelu":
            u = torch.empty((M, w1c.shape[0]), dtype=T, device=x.device)    # gelu'(pre-activation), for the backward epilogue
            h = ops.linear_fwd(xn, w1c, _f32(b1), epilogue=L.EPI_GELU, aux=u)
        else:
            u = None
            h = ops.linear_fwd(xn, w1c, _f32(b1), epilogue=L.EPI_RELU)
        if residual:
            y = ops.linear_fwd(h, w2c, _f32(b2), epilogue=L.EPI_RESIDUAL, residual=x2,
                               out_dtype=torch.float32 if mixed else None)
        else:
            y = ops.linear_fwd(h, w2c, _f32(b2))
        ctx.save_for_backward(x2, g, mean, rstd, xn if prenorm else None, w1c, w2c, u, h)
        ctx.cfg = (act, prenorm, residual, b1 is not None, b2 is not None)
        ctx.xshape = shp
        ctx.sinks = tuple(_sink(t) for t in (ln_w, ln_b, w1, b1, w2, b2))
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, g, mean, rstd, xn, w1c, w2c, u, h = ctx.saved_tensors
        act, prenorm, residual, has_b1, has_b2 = ctx.cfg
        T = h.dtype
        if xn is None:
            xn = x2
        if ctx.mixed:
            dy32 = dy.reshape(x2.shape).contiguous()
            dy2 = _lp_take(dy32, T)
        else:
            dy2 = _as(dy.reshape(x2.shape).contiguous(), T)
        sk = ctx.sinks
        # full-size shapes: each weight gradient's split-K reduce rides in the tail of the data-gradient launch behind it;
        # launch-bound shapes: weight and data gradient of a Linear in one launch
        if act == "gelu":
            dw2, db2, du = _linear_bwd(sk[4], sk[5], dy2, h, w2c, has_b2, epilogue=L.EPI_DGELU, aux=u)
        else:
            dw2, db2, du = _linear_bwd(sk[4], sk[5], dy2, h, w2c, has_b2, epilogue=L.EPI_DRELU, aux=h)
        dw1, db1, dxn = _linear_bwd(sk[2], sk[3], du, xn, w1c, has_b1)
        dg = db = None
        if ctx.mixed:
            dx, dg, db, dx_lp = _ln_bwd(dxn, x2, g, mean, rstd, sk[0], sk[1], dx_add=dy32, dx_dtype=torch.float32, dx_lp=T)
            _lp_put(dx, dx_lp)
        elif prenorm:
            dx, dg, db = _ln_bwd(dxn, x2, g, mean, rstd, sk[0], sk[1], dx_add=dy2 if residual else None)
        else:
            dx = ops.add(dxn, dy2) if residual else dxn
        return dx.view(ctx.xshape), dg, db, dw1, db1, dw2, db2, None, None, None, None, None


def mlp_block(x, ln_w, ln_b, w1, b1, w2, b2, *, act="gelu", prenorm=True, residual=True, eps=1e-5, cdt=None):
    return _MlpBlock.apply(x, ln_w, ln_b, w1, b1, w2, b2, act, prenorm, residual, eps, cdt)


# ---------------------------------------------------------------------------
# Losses (frame_transformer.py:89-90,246-273)
# ---------------------------------------------------------------------------
class _BceLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, target):
        zc = z.contiguous()
        t32 = _f32(target)
        ctx.save_for_backward(zc, t32)
        return ops.bce_logits_fwd(zc, t32).view(())

    @staticmethod
    def backward(ctx, gloss):
        zc, t32 = ctx.saved_tensors
        g = _f32(gloss).reshape(1)
        return ops.bce_logits_bwd(zc, t32, g), None


def bce_with_logits(z: Tensor, target: Tensor) -> Tensor:
    """nn.BCEWithLogitsLoss() (mean)."""
    return _BceLogits.apply(z, target)


class _HeadBce(torch.autograd.Function):
    """loss = mean BCE(Linear(LN2(LN1(x))), target): the classification head of vit.py:97-100,126-128 (LN1 = the temporal
    Transformer's final norm on the pooled row, None when the caller applied it) with the caller's BCEWithLogitsLoss, one
    launch forward (which also leaves every gradient for an upstream gradient of 1) and one launch backward (scaling by the
    incoming gradient and the stores into the gradient buffers)."""

    @staticmethod
    def forward(ctx, x, g1, b1, g2, b2, w, c, target, eps1, eps2, lp_dtype):
        x2 = x.contiguous()
        loss, logits, grads = ops.head_bce_fwd(x2, _f32(g1), _f32(b1), eps1, _f32(g2), _f32(b2), eps2, _f32(w), _f32(c),
                                               _f32(target))
        ctx.grads = grads
        ctx.sinks = tuple(_sink(p) for p in (g1, b1, g2, b2, w, c))
        ctx.shapes = tuple(None if p is None else tuple(p.shape) for p in (g1, b1, g2, b2, w, c))
        ctx.x_dtype, ctx.lp_dtype = x.dtype, lp_dtype
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)             # (no zero tensor for the logits' absent gradient)
        return loss.view(()), logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        G = ctx.grads
        g = _f32(gloss).reshape(1)
        entries, outs, marks = [], [], []
        dx = dx_lp = None
        if ctx.needs_input_grad[0]:
            if ctx.x_dtype == torch.float32:
                dx = torch.empty_like(G["x"])
                # (the block in front wants the row gradient as a 16-bit GEMM operand as well: same launch)
                dx_lp = torch.empty(G["x"].shape, dtype=ctx.lp_dtype, device=dx.device) if ctx.lp_dtype is not None else None
                entries.append((G["x"], dx, False, dx_lp))
            else:
                dx = torch.empty(G["x"].shape, dtype=ctx.x_dtype, device=G["x"].device)
                entries.append((G["x"], None, False, dx))
        for i, name in enumerate(("g1", "b1", "g2", "b2", "w", "c")):
            if ctx.shapes[i] is None or not ctx.needs_input_grad[1 + i]:
                outs.append(None)
                continue
            sk = ctx.sinks[i]
            if sk is not None:
                entries.append((G[name].reshape(-1), sk.buf.view(-1), not sk.fresh, None))
                marks.append(sk)
                outs.append(None)
            else:
                o = torch.empty(ctx.shapes[i], dtype=torch.float32, device=g.device)
                entries.append((G[name].reshape(-1), o.view(-1), False, None))
                outs.append(o)
        ops.scaled_emit_group(g, entries)
        for sk in marks:
            sk.mark_written()
        if dx_lp is not None:
            _lp_put(dx, dx_lp)
        return (dx, *outs, None, None, None, None)


def head_bce_supported(x: Tensor, w: Tensor) -> bool:
    return x.dim() == 2 and x.is_cuda and ops.head_bce_supported(x.shape[0], x.shape[1], w.shape[0])


def head_bce(x, ln1, ln2, linear, target, *, lp_dtype=None):
    """-> (loss, logits).  ln1: LayerNorm module or None; ln2: LayerNorm; linear: Linear; x [rows, d] fp32 or 16-bit."""
    g1, b1, e1 = (ln1.weight, ln1.bias, ln1.eps) if ln1 is not None else (None, None, 0.0)
    return _HeadBce.apply(x, g1, b1, ln2.weight, ln2.bias, linear.weight, linear.bias, target, e1, ln2.eps, lp_dtype)


class _CeArgmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher):
        s = student.contiguous()
        t = _as(teacher.detach().contiguous(), s.dtype)
        ctx.save_for_backward(s, t)
        return ops.ce_argmax_fwd(s, t).view(())

    @staticmethod
    def backward(ctx, gloss):
        s, t = ctx.saved_tensors
        return ops.ce_argmax_bwd(s, t, _f32(gloss).reshape(1)), None


def cross_entropy_argmax(student: Tensor, teacher: Tensor) -> Tensor:
    """CrossEntropyLoss(student, argmax(teacher, -1)) -- the hard-label distillation term."""
    return _CeArgmax.apply(student, teacher)


# ---------------------------------------------------------------------------
# Sinusoidal positional encoding add and row concatenation (frame_transformer.py:19-34,
# transformer.py:74-82)
# ---------------------------------------------------------------------------
class _AddRowTable(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, rows_per_entry):
        return ops.add_rowtable(x.contiguous(), _f32(table).contiguous(), rows_per_entry)

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None


def add_positional_table(x: Tensor, pe: Tensor) -> Tensor:
    """x [L, B, E] seq-first; pe [max_len, 1, E] buffer: x + pe[:L] (broadcast over B)."""
    Ln, B = x.shape[0], x.shape[1]
    if pe.shape[0] < Ln:
        raise ValueError(f"positional table has {pe.shape[0]} rows but the sequence has {Ln} "
                         "(frame_transformer.py:92-93: max_len must cover the sequence)")
    return _AddRowTable.apply(x, pe[:Ln].reshape(Ln, -1), B)


class _ConcatRows(torch.autograd.Function):
    """cat((a, b), dim=0) for contiguous tensors sharing trailing dims (device copies only)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), dtype=b.dtype, device=b.device)
        ops.copy_(out[: a.shape[0]], a)
        ops.copy_(out[a.shape[0]:], b)
        ctx.na, ctx.adt = a.shape[0], a.dtype
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        da = torch.empty(dy[: ctx.na].shape, dtype=ctx.adt, device=dy.device)
        ops.copy_(da, dy[: ctx.na])
        db = torch.empty_like(dy[ctx.na:])
        ops.copy_(db, dy[ctx.na:])
        return da, db


def concat_rows(a: Tensor, b: Tensor) -> Tensor:
    return _ConcatRows.apply(a, b)


class _MeanRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.L, ctx.scale = x.shape[1], scale
        return ops.mean_rows_fwd(x, scale)

    @staticmethod
    def backward(ctx, dy):
        return ops.mean_rows_bwd(dy, ctx.L, ctx.scale), None


def mean_rows(x: Tensor) -> Tensor:
    """x [B, L, d] -> mean over L (``x.mean(dim=1)``, vit.py:126; global AvgPool2d, TPN.py:6,20,33)."""
    return _MeanRows.apply(x, None)


def sum_rows(x: Tensor) -> Tensor:
    """x [B, L, d] -> sum over L (``sum_group``, TPN.py:64-72)."""
    return _MeanRows.apply(x, 1.0)


def sigmoid(x: Tensor) -> Tensor:
    return _Act.apply(x, ops.ACT_SIGMOID)


class _ConcatCols(torch.autograd.Function):
    """cat(tensors, dim=-1) of [rows, d_i] matrices (TPN.py:58), device copies only."""

    @staticmethod
    def forward(ctx, *xs):
        rows = xs[0].shape[0]
        widths = [x.shape[1] for x in xs]
        out = torch.empty((rows, sum(widths)), dtype=xs[0].dtype, device=xs[0].device)
        off = 0
        for x, w in zip(xs, widths):
            ops.copy2d(x.contiguous(), out[:, off:], rows, w, w, out.shape[1])
            off += w
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        rows, tot = dy.shape
        outs, off = [], 0
        for w in ctx.widths:
            g = torch.empty((rows, w), dtype=dy.dtype, device=dy.device)
            ops.copy2d(dy[:, off:], g, rows, w, tot, w)
            outs.append(g)
            off += w
        return tuple(outs)


def concat_cols(*xs: Tensor) -> Tensor:
    return _ConcatCols.apply(*xs)


class _ScaleF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        ops.axpby_f32_(out.view(-1), x.contiguous().view(-1), alpha, 0.0)
        return out

    @staticmethod
    def backward(ctx, dy):
        out = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
        ops.axpby_f32_(out.view(-1), dy.contiguous().view(-1), ctx.alpha, 0.0)
        return out, None


def scale_f32(x: Tensor, alpha: float) -> Tensor:
    """alpha * x as fp32 (the average of the three scale predictions, TPN.py:112)."""
    return _ScaleF32.apply(x, alpha)


class _Permute021(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.permute_021(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.permute_021(dy)


def to_seq_first(x: Tensor) -> Tensor:
    """'b s d -> s b d' (frame_transformer.py:205); also its own inverse."""
    return _Permute021.apply(x)


class _SelectRow(torch.autograd.Function):
    """seq [S, B, E] -> seq[i]  ([B, E]); the gradient is zero elsewhere."""

    @staticmethod
    def forward(ctx, seq, i):
        seq = seq.contiguous()
        ctx.shape, ctx.i = tuple(seq.shape), i
        out = torch.empty(seq.shape[1:], dtype=seq.dtype, device=seq.device)
        ops.copy_(out, seq[i])
        return out

    @staticmethod
    def backward(ctx, dy):
        dx = ops.zeros(ctx.shape, dy.dtype, dy.device)
        ops.copy_(dx[ctx.i], dy.contiguous())
        return dx, None


def select_seq_first_row(seq: Tensor, i: int) -> Tensor:
    return _SelectRow.apply(seq, i)


class _ClsConcat(torch.autograd.Function):
    """Per-sample cat((cls, data[b]), dim=0): data [B, S, X], cls [X] (learnable, broadcast)
    -> [B, S+1, X]  (frame_transformer.py:194-197, 213-217)."""

    @staticmethod
    def forward(ctx, data, cls):
        data = data.contiguous()
        B, S, X = data.shape
        c = _as(cls.detach().reshape(-1).contiguous(), data.dtype)
        out = torch.empty((B, S + 1, X), dtype=data.dtype, device=data.device)
        ops.copy2d(c, out, B, X, 0, (S + 1) * X)                       # broadcast row 0 of every sample
        ops.copy2d(data, out[:, 1:], B, S * X, S * X, (S + 1) * X)
        ctx.dims = (B, S, X)
        ctx.cls_shape, ctx.data_grad = tuple(cls.shape), data.requires_grad
        ctx.sink = _sink(cls)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, S, X = ctx.dims
        dout = dout.contiguous()
        ddata = None
        if ctx.needs_input_grad[0]:
            ddata = torch.empty((B, S, X), dtype=dout.dtype, device=dout.device)
            ops.copy2d(dout[:, 1:], ddata, B, S * X, (S + 1) * X, S * X)
        sk = ctx.sink
        if sk is not None:
            ops.rows_sum(dout, (S + 1) * X, B, X, out=sk.buf.view(-1), accumulate=not sk.fresh)
            sk.mark_written()
            return ddata, None
        dcls = ops.rows_sum(dout, (S + 1) * X, B, X).view(ctx.cls_shape)
        return ddata, dcls


def cls_concat(data: Tensor, cls: Tensor) -> Tensor:
    return _ClsConcat.apply(data, cls)


# ---------------------------------------------------------------------------
# Per-frame CNN encoder blocks (src/models/custom_resnet.py): conv -> BN -> (+res) -> ReLU
# on NHWC feature maps stored as [N*H*W, C] matrices.
# ---------------------------------------------------------------------------
def _kpad(K: int, dtype: torch.dtype) -> int:
    """Reduction length padded so that the LDS-DMA MFMA kernel is eligible (K % 64 == 0)."""
    return K if dtype == torch.float32 else (K + 63) // 64 * 64


# Keep the im2col column matrix of every convolution for its weight-gradient GEMM instead of re-gathering it in
# backward: ~9x the activation bytes per 3x3 layer (8 GB for ResNet-18 on 256 frames of 224^2) -- affordable in
# 288 GB of HBM and one gather pass per layer cheaper.  Set False to trade the memory back.
SAVE_CONV_COLUMNS = True
# Implicit-GEMM convolution (gather fused into the GEMM operand DMA) for the forward pass and, for stride-1
# convolutions, the data gradient, wherever the geometry allows (16-bit dtype, Cin a multiple of 32/64).
IMPLICIT_CONV = True
# BatchNorm batch statistics from the implicit convolution's GEMM epilogue (fp32 accumulators) instead of a separate pass
# over the stored output
FUSE_BN_STATS = True
GENERAL_TAPS = os.environ.get("DVT_GENERAL_TAPS", "1") != "0"   # implicit kernels for any C % 8 == 0 (per-lane taps), K not padded to 64
HALO_CONV = True          # 64 -> 64 channel 3x3 / 1 / 1 convolutions from an LDS-resident halo patch (dvt_conv3x3_c64)
# data gradient of STRIDED convolutions as one implicit launch per parity class of input pixels (1 / 2 / 2 / 4 taps of a
# 3 x 3 / 2 filter) scattering into the full-size gradient, instead of a dcol GEMM + col2im pass
STRIDED_IMPLICIT = os.environ.get("DVT_STRIDED_IMPLICIT", "1") != "0"
# ... for maps of at least this many input pixels: below it the four class launches are latency-bound (17 - 21 us each at
# 12 k rows per class, `gpurun_out/r5_pyr_strided_order.txt`) and the dcol GEMM + col2im pair is as fast
STRIDED_IMPLICIT_MIN_PIXELS = int(os.environ.get("DVT_STRIDED_IMPLICIT_MIN_PIXELS", "150000"))


def _packed_weight(w: Tensor, kind: int, cout_l: int, cin_l: int, kh: int, kw: int, cout_p: int, cin_p: int, ld: int,
                   dtype: torch.dtype) -> Tensor:
    """The GEMM operand form of a convolution weight f32 [cout_l, cin_l, (1,) kh, kw]: kind 0 = forward [cout_p, ld] (column
    (ki*kw + kj) * cin_p + ci), kind 1 = data gradient [cin_p, kh*kw*cout_p] (rotated taps, transposed channels); channels
    beyond the parameter's own are zero (channel-padded layers).  A parameter that lives in a ``dp.FlatParameters`` store keeps
    its packed forms there and ALL of them are refreshed by one launch per optimizer step (the weights only change there);
    any other tensor is packed on the spot."""
    def make():
        src = w.detach().contiguous()
        shape = (cout_p, ld) if kind == 0 else (cin_p, kh * kw * cout_p)
        dst = torch.empty(shape, dtype=dtype, device=w.device)
        return dst, (src, dst, cout_l, cin_l, kh, kw, cout_p, cin_p, ld, kind)
    sink = _sink(w)
    if sink is None or w.dtype != torch.float32:
        dst, entry = make()
        ops.conv_weight_pack_group([entry])
        return dst
    return sink.owner.packed_weight((sink.index, kind, cout_p, cin_p, ld, dtype), make, w._version)


def _packed_class_weight(w: Tensor, cout_l: int, cin_l: int, kh: int, kw: int, cout_p: int, cin_p: int, cls, dtype) -> Tensor:
    """The data-gradient operand of one parity class of a strided convolution (dvt_pack_entry kind 2): [cin_p, nth*ntw*cout_p]
    with the class's taps only; cached and refreshed like the other packed forms (`_packed_weight`)."""
    sh, sw, rh, rw = cls
    ntaps = ((kh - rh + sh - 1) // sh) * ((kw - rw + sw - 1) // sw)

    def make():
        src = w.detach().contiguous()
        dst = torch.empty((cin_p, ntaps * cout_p), dtype=dtype, device=w.device)
        return dst, (src, dst, cout_l, cin_l, kh, kw, cout_p, cin_p, 0, 2, (sh, sw, rh, rw))
    sink = _sink(w)
    if sink is None or w.dtype != torch.float32:
        dst, entry = make()
        ops.conv_weight_pack_group([entry])
        return dst
    return sink.owner.packed_weight((sink.index, 2, cout_p, cin_p, (sh, sw, rh, rw), dtype), make, w._version)


# the 7x7 / 2 stem from an LDS halo patch with the weights in registers (csrc/conv_stem.hip) instead of the implicit gather
STEM_HALO = os.environ.get("DVT_STEM_HALO", "1") != "0"
# weight gradient of the 64 -> 64 3x3 layers from LDS halo patches (csrc/conv3x3_wgrad.hip) instead of the implicit gather
HALO_WGRAD = os.environ.get("DVT_HALO_WGRAD", "1") != "0"
HALO_WGRAD_COUT = (64, 144) if os.environ.get("DVT_HALO_WGRAD_WIDE", "1") != "0" else (64,)
# weight gradient of the (3, 1) temporal convolutions 144 -> 64 from LDS sliding windows (csrc/conv3x1_wgrad.hip)
WINDOW_WGRAD = os.environ.get("DVT_WINDOW_WGRAD", "1") != "0"
# ... and their forward (csrc/conv3x1_fwd.hip)
WINDOW_FWD = os.environ.get("DVT_WINDOW_FWD", "1") != "0"
# "virtual" BatchNorm between the two halves of such a pair: the spatial half leaves its output z and hands (mean, invstd,
# gamma, beta) on; the temporal half's window kernels form relu(z * s + t) in LDS -- the normalised 144-plane activation is
# never written (models/video_resnet.py decides per pair; needs both window kernels)
WINDOW_VIRTUAL_BN = os.environ.get("DVT_WINDOW_VIRTUAL_BN", "1") != "0"
# the backward of that virtual BatchNorm inside the temporal half's data gradient (ops.conv3x1_stream_bn_bwd: the 144-plane
# gradient is computed twice and never stored) instead of a stored gradient + two BatchNorm passes over it
FUSED_MID_BN_BWD = os.environ.get("DVT_FUSED_MID_BN_BWD", "1") != "0"

class _ConvBnAct(torch.autograd.Function):
    """y = relu?( BN(conv(x)) (+ residual) ).  x: NHWC matrix [N*H*W, Cin], or the raw NCHW
    clip frames [N, Cin, H, W] for the stem.  Returns the NHWC matrix [N*Ho*Wo, Cout].

    Channel padding (``cpad`` > 0): Cout is rounded up to a multiple of ``cpad`` and x may carry more channels
    than ``w`` has input planes (a padded predecessor); weights / BatchNorm vectors are zero-extended on the fly, so
    the padded output channels are exactly zero and every kernel sees MFMA-friendly widths (R(2+1)D mid planes
    45 / 230 / 460 / 921).  Parameters and their gradients keep the reference shapes."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, residual, run_mean, run_var, geom, relu, training, momentum, eps, dtype,
                cpad=0, dx_frames=None, pool=False, fork=None, in_affine=None, defer_apply=False, link=None):
        # in_affine = (mean, invstd, gamma32, beta32, c_valid, relu): x is the OUTPUT z of the convolution in front and the
        # BatchNorm (+ ReLU) between the two layers is applied inside this layer's kernels (csrc/conv3x1_window.h); only the
        # window kernels of the (3, 1) temporal 144 -> 64 layers take it.  defer_apply: the mirror image on the producer's
        # side -- no bn_apply_fwd launch; the Function returns (z, mean, invstd) and the caller hands the affine on.
        ctx.in_affine = in_affine
        # link (defer_apply side; the consumer finds it as in_affine[6]): a dict shared by the two Functions of a virtual pair.
        # The consumer's backward may run THIS layer's BatchNorm backward inside its own data gradient (FUSED_MID_BN_BWD): it
        # then finds the gradient sinks of gamma / beta here, leaves dgamma / dbeta and sets "done", and this layer's backward
        # takes the incoming gradient as dz.
        ctx.link = link
        # fork: the layer's input has a second consumer, the block's shortcut (custom_resnet.py:38-54), handed out by THIS
        # Function as a second output so that its gradient arrives here and joins the data gradient inside the kernel that
        # produces it (residual epilogue of the implicit / halo convolution, second operand of col2im) instead of in an add
        # kernel behind it.  "alias": the second output is x itself; an int s: x subsampled with stride s, i.e. the input of
        # a strided 1x1 downsample convolution, whose gradient then comes back COMPACT (no zero-filled full-size map).
        ctx.fork = fork
        # pool: the layer is followed by MaxPool2d(3, 2, 1) (the ResNet stem, custom_resnet.py:100-105): BatchNorm, ReLU and
        # the pooling run as one pass over the convolution output and the pooled map is returned; backward gathers the
        # layer's gradient from the pooled gradient inside the BatchNorm backward (neither full-resolution map exists).
        # dx_frames (NCHW stem only): [(first_frame, count), ...] -- the only input frames whose gradient is consumed (the
        # learnable pixel-space CLS chunk of each sample, frame_transformer.py:105,195); other frames get zeros unseen.
        ctx.dx_frames = dx_frames
        N, Cin, H, W, k, stride, pad, nchw = geom
        Cout_l, Cin_l = w.shape[0], w.shape[1]
        Cout = (Cout_l + cpad - 1) // cpad * cpad if cpad else Cout_l
        (kh, kw), (sh, sw) = ops._pair(k), ops._pair(stride)
        Ho, Wo = ops.conv_out_hw(H, W, k, stride, pad)
        xc = x.contiguous()
        ctx.pair = pair = None
        ctx.trim = 0
        # (frames that need a gradient: theirs is an explicit dcol GEMM + NCHW col2im over the rows of the dx_frames hint --
        # all frames without one -- and everything else about the layer stays implicit)
        stem8 = (IMPLICIT_CONV and nchw and Cin <= 8 and dtype in (torch.bfloat16, torch.float16) and Cout % 8 == 0
                 and (N * Ho * Wo) % 32 == 0)
        ctx.geom0 = None
        if stem8:
            ctx.geom0 = (N, Cin, H, W, k, stride, pad)
            ctx.w_stem = w if x.requires_grad else None
            if x.requires_grad and not dx_frames:
                ctx.dx_frames = [(0, N)]
            # The stem (custom_resnet.py:100: 7x7 / 2 on the 3-channel frames) as an implicit GEMM: the frames become an
            # NHWC map with the channels zero-extended to 8 (one 16-byte chunk per pixel and tap), the weights carry the
            # matching zero planes, and from here on it is an ordinary NHWC convolution with Cin = 8 -- no column matrix
            # (0.94 GB at 256 frames of 224^2) in forward or in the weight gradient.
            (ph, pw) = ops._pair(pad)
            off = pw & 1
            kwp = (kw + off + 1) // 2                      # pairs that cover the taps: (-1,0) (1,2) (3,4) (5,6) for 7 / pad 3
            trim = (W // 2 + (pw + off) - kwp + 1) - Wo if (sw == 2 and W % 2 == 0) else -1
            if Cin <= 4 and Cin == Cin_l and trim >= 0:
                # Stride 2 and <= 4 channels: two horizontally adjacent pixels share a 16-byte chunk (Cpad = 4), and in that
                # [N, H, W/2, 8] view the stem is a (kh, kwp) convolution of stride (sh, 1) over pixel pairs whose weights
                # are the stem's, re-laid by dvt_conv_weight_pairs (zero taps where a pair sticks out of the kernel): 28
                # gathered chunks per output pixel instead of 49, K = 224 instead of 392 (one tile column of the weight
                # gradient instead of two).  Its symmetric padding of (pw + off) / 2 pair columns yields `trim` output
                # columns too many on the right (dvt_conv_desc.trim_w).  From here on it IS that convolution.
                ctx.pair = pair = (kh, kw, pw, kwp)
                ctx.trim = trim
                xc = ops.nchw_to_nhwc_pad(xc.view(N, Cin, H, W), dtype, 4)
                W, k, stride, pad = W // 2, (kh, kwp), (sh, 1), (ph, (pw + off) // 2)
                (kh, kw), (sh, sw) = k, stride
            else:
                xc = ops.nchw_to_nhwc_pad(xc.view(N, Cin, H, W), dtype, 8)
            Cin, nchw = 8, False
            geom = (N, Cin, H, W, k, stride, pad, nchw)
        padded = Cout != Cout_l or Cin != Cin_l
        if Cin < Cin_l:
            raise ValueError("input has fewer channels than the convolution weight")
        K = kh * kw * Cin
        direct = (kh == 1 and kw == 1 and sh == 1 and sw == 1 and not nchw and K % 8 == 0 and x.dtype == dtype)
        # (a 1x1 / stride-1 layer the implicit kernels accept goes through them as well: the BatchNorm statistics come out of
        # their epilogue and the weight gradient's reduce rides in the data gradient, like every other layer)
        if direct and IMPLICIT_CONV and dtype in (torch.bfloat16, torch.float16) and Cout % 8 == 0 and \
                Cin % (32 if Cout <= 128 else 64) == 0 and (N * H * W) % 64 == 0:
            direct = False
        # C % 8 == 0 but not a whole number of k-tiles per filter tap (144 mid planes of R(2+1)D-18): the implicit kernels
        # derive every lane's tap themselves and take K rounded up to THEIR k-tile (zero weight columns behind kh*kw*C)
        # (and K itself, not K rounded up to 64, where the taps are whole k-tiles: 288 mid planes x 3 temporal taps = 864)
        gen = (IMPLICIT_CONV and GENERAL_TAPS and not direct and not nchw and not stem8 and dtype in (torch.bfloat16, torch.float16)
               and xc.dtype == dtype and xc.is_cuda and Cin % 8 == 0 and Cout % 8 == 0)
        ld = K if direct else (ops.conv2d_implicit_k(Cin, Cout, k) if (stem8 or gen) else _kpad(K, dtype))
        if pair is not None:
            kh_o, kw_o, pw_o, _ = pair
            w4 = w.reshape(Cout_l, Cin_l, kh_o * kw_o)
            if Cout != Cout_l:
                w4 = ops.pad3_f32(w4, Cout_l, Cin_l, kh_o * kw_o, Cout, Cin_l)
            w4 = ops.conv_weight_pairs(w4, Cout, Cin_l, kh_o, kw_o, pw_o, kwp)          # [Cout, 8, kh, kwp]
            wp = ops.conv_weight_pack(w4, ld, dtype)
        else:                                      # packed once per optimizer step for all layers (zero extension included)
            wp = _packed_weight(w, 0, Cout_l, Cin_l, kh, kw, Cout, Cin, ld, dtype)
        implicit = (IMPLICIT_CONV and not direct and not nchw and (ld == K or stem8 or gen) and xc.dtype == dtype and
                    ops.conv2d_implicit_supported(xc, wp, N, Cin, H, W, Cout, k, stride, pad, ctx.trim))
        if gen and not implicit:                   # (misaligned operands: the explicit path and its 64-padded columns)
            ld = _kpad(K, dtype)
            wp = _packed_weight(w, 0, Cout_l, Cin_l, kh, kw, Cout, Cin, ld, dtype)
        if ctx.trim and not implicit:
            raise RuntimeError("the pixel-pair stem needs the implicit convolution kernels")
        stats_partial = None
        halo = (HALO_CONV and implicit and Cin == 64 and Cout == 64 and (kh, kw) == (3, 3) and (sh, sw) == (1, 1)
                and ops._pair(pad) == (1, 1) and ops.conv3x3_c64_supported(xc, wp, N, H, W))
        stream = (HALO_CONV and implicit and not halo and ld == K and (kh, kw) == (3, 3) and (sh, sw) == (1, 1)
                  and ops._pair(pad) == (1, 1) and ops.conv3x3_stream_supported(xc, wp, N, H, W, Cin, Cout))
        # (the 64 -> 64 form also takes a zero-extended input: the stem's 45 mid planes stored as 64)
        window = (WINDOW_FWD and HALO_CONV and implicit and (not padded or (Cin == 64 and Cout == Cout_l == 64))
                  and (kh, kw) == (3, 1) and (sh, sw) == (1, 1)
                  and ops._pair(pad) == (1, 0) and ops.conv3x1_fwd_supported(xc, wp, N, H, W, Cin, Cout))
        if in_affine is not None and not window:
            raise RuntimeError("in_affine (virtual BatchNorm in front of the layer) needs the window kernels of the (3, 1) "
                               "temporal 144 -> 64 convolution")
        # the pixel-pair stem (7x7 / 2 / 3 on 3-channel frames, 64 output channels incl. zero extension) from an LDS halo patch
        stem_halo = (STEM_HALO and HALO_CONV and implicit and pair is not None and Cout == 64 and (kh, kw) == (7, 4)
                     and (sh, sw) == (2, 1) and ops._pair(pad) == (3, 2) and ctx.trim == 1
                     and ops.conv_stem7_supported(xc, wp, N, H, W))
        if stem_halo:
            col = None
            halo = stream = window = False
            if training and FUSE_BN_STATS:
                z, stats_partial, stats_parts = ops.conv_stem7(xc, wp, N, H, W, want_stats=True)
            else:
                z = ops.conv_stem7(xc, wp, N, H, W)
        elif window:                                # temporal half of R(2+1)D-18's layer-1 pairs: a pixel segment over all frames in LDS
            col = None
            halo = stream = False
            if training and FUSE_BN_STATS:
                z, stats_partial, stats_parts = ops.conv3x1_fwd(xc, wp, N, H, W, want_stats=True, affine=in_affine)
            else:
                z = ops.conv3x1_fwd(xc, wp, N, H, W, affine=in_affine)
        elif stream:                              # layer 1 of R(2+1)D-18, 64 -> 144: halo patch, weights streamed through LDS
            col = None
            if training and FUSE_BN_STATS:
                z, stats_partial, stats_parts = ops.conv3x3_stream(xc, wp, N, H, W, Cin, Cout, want_stats=True)
            else:
                z = ops.conv3x3_stream(xc, wp, N, H, W, Cin, Cout)
        elif halo:                                # layer 1 of ResNet-18: LDS-resident halo patch instead of nine gathers
            col = None
            if training and FUSE_BN_STATS:
                z, stats_partial, stats_parts = ops.conv3x3_c64(xc, wp, N, H, W, want_stats=True)
            else:
                z = ops.conv3x3_c64(xc, wp, N, H, W)
        elif implicit:
            col = None
            if training and FUSE_BN_STATS:       # column sums for the BatchNorm come out of the GEMM epilogue
                z, stats_partial, stats_parts = ops.conv2d_implicit(xc, wp, N, Cin, H, W, Cout, k, stride, pad, want_stats=True,
                                                                    trim_w=ctx.trim)
            else:
                z = ops.conv2d_implicit(xc, wp, N, Cin, H, W, Cout, k, stride, pad, trim_w=ctx.trim)
        else:
            col = xc if direct else ops.im2col(xc, nchw, N, Cin, H, W, k, stride, pad, ld, dtype)
            z = ops.linear_fwd(col, wp)                                 # [N*Ho*Wo, Cout]
        # BatchNorm vectors keep the parameter's own length; the kernels take it as c_valid and treat the padded channels as
        # gamma = beta = 0 (no padded copies, no slices of the statistics or of dgamma / dbeta)
        g32, b32 = _f32(gamma), _f32(beta)
        rm, rv = run_mean, run_var
        cval = Cout_l if Cout != Cout_l else 0
        if training:
            if stats_partial is not None:
                mean, invstd = ops.bn_stats_from_partials(stats_partial, stats_parts, z.shape[0], Cout, rm, rv, eps, momentum,
                                                          c_valid=cval)
            else:
                mean, invstd = ops.bn_stats(z, rm, rv, eps, momentum, c_valid=cval)
        else:
            mean, invstd = rm.detach().float(), ops.bn_eval_invstd(rv.detach().float(), eps)
            if cval:                               # (mean / invstd are internal arrays of the padded width)
                mean = ops.pad3_f32(mean, Cout_l, 1, 1, Cout, 1).view(-1)
                invstd = ops.pad3_f32(invstd, Cout_l, 1, 1, Cout, 1).view(-1)
        res = None if residual is None else residual.contiguous()
        if res is not None and res.shape[1] != Cout:
            raise ValueError("residual width must equal the (padded) output width")
        if defer_apply and (res is not None or pool or dtype not in (torch.bfloat16, torch.float16)):
            raise RuntimeError("defer_apply is for plain 16-bit conv -> BatchNorm (-> ReLU) layers")
        pooled = bool(pool) and res is None and Cout % 8 == 0
        if pool and not pooled:
            raise ValueError("pool=True needs a layer without a residual branch and a multiple of 8 output channels")
        pidx = None
        # a ReLU layer with a residual branch cannot recompute its mask from z alone: the forward leaves one bit per element
        # (1/16 of re-reading y in both passes of the BatchNorm backward)
        rmask = None
        if pooled:
            if cval:
                raise ValueError("pool=True is for layers without channel padding")
            y, pidx = ops.bn_relu_maxpool_fwd(z, mean, invstd, g32, b32, N, Cout, Ho, Wo, relu)
        elif defer_apply:
            y = z                                 # the consumer forms relu(z * s + t) in its staged window
        elif relu and res is not None and Cout % 8 == 0 and any(ctx.needs_input_grad):
            y, rmask = ops.bn_apply_fwd(z, mean, invstd, g32, b32, res, relu, want_mask=True, c_valid=cval)
        else:
            y = ops.bn_apply_fwd(z, mean, invstd, g32, b32, res, relu, c_valid=cval)
        # weight gradient straight from x (column matrix gathered inside the GEMM): nothing to keep but x
        wg_implicit = (IMPLICIT_CONV and not direct and not nchw and xc.dtype == dtype and
                       ops.conv2d_implicit_wgrad_supported(xc, z, N, Cin, H, W, Cout, k, stride, pad, ctx.trim))
        if ctx.trim and not wg_implicit:
            raise RuntimeError("the pixel-pair stem needs the implicit weight-gradient kernel")
        keep_col = SAVE_CONV_COLUMNS and not direct and col is not None and not wg_implicit
        # a ReLU layer without a residual branch recomputes its mask from z in backward (saves two passes over y)
        keep_y = relu and residual is not None
        ctx.save_for_backward(None if keep_col else xc, wp, z, y if (keep_y and rmask is None) else None, mean, invstd, g32,
                              col if keep_col else None, b32 if ((relu and not keep_y) or pooled) else None, pidx, rmask)
        ctx.cfg = (geom, Cout, ld, direct, relu, training, residual is not None, tuple(w.shape), dtype)
        ctx.sinks = (_sink(w), _sink(gamma), _sink(beta))
        if link is not None:
            link["sinks"], link["training"], link["done"], link["grads"] = (ctx.sinks[1], ctx.sinks[2]), training, False, (None, None)
        ctx.x_needs = x.requires_grad
        ctx.x_shape, ctx.x_dtype = tuple(x.shape), x.dtype
        ctx.implicit = implicit
        ctx.w_ref = w if pair is None else None           # the parameter itself: backward asks for its data-gradient form
        ctx.w4 = w4.detach() if (implicit and pair is not None) else None
        ctx.wg_implicit = wg_implicit
        ctx.logical = (Cout_l, Cin_l, padded)
        ctx.defer_apply = bool(defer_apply)
        tail = ()
        if defer_apply:                            # (mean, invstd) ride behind the data outputs
            ctx.mark_non_differentiable(mean, invstd)
            # (autograd otherwise hands backward two zero-filled [C] "gradients" for them: two fill launches per pair)
            ctx.set_materialize_grads(False)
            y, tail = y.view(y.shape), (mean, invstd)
        if fork is None:
            return (y,) + tail if tail else y
        if nchw or stem8:
            raise ValueError("fork is for NHWC feature maps (residual blocks), not the stem")
        if fork == "alias":
            return (y, x.view(x.shape)) + tail
        return (y, ops.im2col(xc, False, N, Cin, H, W, 1, int(fork), 0, Cin, xc.dtype)) + tail

    @staticmethod
    def backward(ctx, dy, *more):
        if _ln_pending_sinks:
            # backward has left the transformer for the per-frame CNN encoder: whatever LayerNorm dgamma / dbeta reduces are
            # still deferred (the launch-bound zone's, a token LayerNorm of the front-end) are performed NOW, not at the end of
            # the step -- under data parallelism their gradient buckets then exchange beside the encoder's whole backward
            # instead of after it (frametransformer at one rank: 68 MB of buckets moved from -0.09 .. -0.01 ms before the end
            # of backward to its start, bench.py --force-dist)
            ln_flush()
        dshort = more[0] if (ctx.fork is not None and more) else None     # (behind it, under defer_apply: mean and invstd's)
        if dy is None or (ctx.fork is not None and dshort is None):       # (unmaterialised gradients of a defer_apply layer)
            raise RuntimeError("_ConvBnAct.backward: an output of a deferred-BatchNorm layer received no gradient")
        xc, wp, z, y, mean, invstd, g32, col, b32, pidx, rmask = ctx.saved_tensors
        geom, Cout, ld, direct, relu, training, has_res, wshape, dtype = ctx.cfg
        N, Cin, H, W, k, stride, pad, nchw = geom
        Cout_l, Cin_l, padded = ctx.logical
        sw, sg, sb = ctx.sinks
        dy = _as(dy.contiguous(), z.dtype)
        Ho, Wo = ops.conv_out_hw(H, W, k, stride, pad)
        Wo -= ctx.trim

        cval = Cout_l if Cout != Cout_l else 0           # channel-padded layer: dgamma / dbeta have the parameter's own length

        def bn_backward(**kw):
            if pidx is not None:       # dy is the pooled gradient
                dz_, dg_, db_ = ops.bn_bwd_pooled(dy, pidx, z, mean, invstd, g32, b32, N, Ho, Wo, relu, training, **kw)
                return dz_, None, dg_, db_
            return ops.bn_bwd(dy, z, y, mean, invstd, g32, relu, training, has_res, beta=b32, mask=rmask, c_valid=cval, **kw)

        if ctx.link is not None and ctx.link.get("done"):
            # the consumer's data gradient already went through this layer's BatchNorm backward (and emitted dgamma / dbeta)
            dz, dres = dy, None
            dgam, dbet = ctx.link["grads"]
            ctx.link["done"] = False
        elif sg is not None and sb is not None and sg.fresh != sb.fresh:
            dz, dres, dgam, dbet = bn_backward()
            _emit_into(sg, dgam); _emit_into(sb, dbet)
            dgam = dbet = None
        elif sg is not None and sb is not None:
            dz, dres, _, _ = bn_backward(dgamma=sg.buf.view(-1), dbeta=sb.buf.view(-1), accumulate=not sg.fresh)
            sg.mark_written(); sb.mark_written()
            dgam = dbet = None
        else:
            dz, dres, dgam, dbet = bn_backward()
        (kh, kw) = ops._pair(k)
        w4 = (Cout, Cin, kh, kw)
        # the common case: the split-K reduce scatters straight into the parameter's own gradient layout (no packed dWt, no
        # scatter launch; channel-padded layers: only the entries the parameter has); the pixel-pair stem post-processes the
        # packed form instead
        direct_dw = ctx.wg_implicit and ctx.pair is None
        if direct_dw:
            dw_master = sw.buf.view(wshape) if sw is not None else torch.empty(wshape, dtype=torch.float32, device=dz.device)
            acc_w = (not sw.fresh) if sw is not None else False
            if (HALO_WGRAD and HALO_CONV and Cin == 64 and Cout in HALO_WGRAD_COUT and not padded and (kh, kw) == (3, 3)
                    and ops._pair(stride) == (1, 1) and ops._pair(pad) == (1, 1)
                    and ops.conv3x3_c64_wgrad_supported(xc, dz, N, H, W, Cout)):
                # layer 1 of ResNet-18 (and, in 64-channel groups of dz, of R(2+1)D-18: 64 -> 144): input patch and gradient
                # tile staged once per R rows, the nine taps read from LDS
                pend = ops.conv3x3_c64_wgrad(xc, dz, N, H, W, dw_master, accumulate=acc_w, defer_reduce=True, Cout=Cout)
            elif ((WINDOW_WGRAD or ctx.in_affine is not None) and HALO_CONV and (kh, kw) == (3, 1) and not padded and ops._pair(stride) == (1, 1)
                    and ops._pair(pad) == (1, 0) and ops.conv3x1_wgrad_supported(xc, dz, N, H, W, Cin, Cout)):
                # the temporal half of R(2+1)D-18's layer-1 pairs (144 mid planes -> 64): a segment of pixels over all frames
                # staged once, the three taps read from LDS (the implicit form gathered x once per tap)
                pend = ops.conv3x1_wgrad(xc, dz, N, H, W, dw_master, accumulate=acc_w, defer_reduce=True, affine=ctx.in_affine)
            else:
                xw = xc
                if ctx.in_affine is not None:
                    # the forward kept the BatchNorm in front virtual, and the window weight-gradient kernel refuses what it
                    # is handed now (a switch flipped between forward and backward, a gradient the kernel cannot take):
                    # materialise relu(bn(z)) once -- what the forward skipped -- and take the implicit kernel
                    am, ai, ag, ab, acv, arelu = ctx.in_affine[:6]
                    xw = ops.bn_apply_fwd(xc.contiguous(), am, ai, ag, ab, None, bool(arelu), c_valid=int(acv))
                _, pend = ops.conv2d_implicit_wgrad(xw, dz, N, Cin, H, W, Cout, k, stride, pad, ctx.trim, defer_reduce=True,
                                                    master=dw_master, accumulate=acc_w, logical=(Cout_l, Cin_l))
            unpack = dwp = None
        elif ctx.wg_implicit:
            dwt, pend = ops.conv2d_implicit_wgrad(xc, dz, N, Cin, H, W, Cout, k, stride, pad, ctx.trim,
                                                  defer_reduce=True)                        # [kh*kw*Cin, Cout] fp32
            unpack, dwp = ops.conv_weight_unpack_grad_t, dwt
        else:
            if col is None:
                col = xc if direct else ops.im2col(xc, nchw, N, Cin, H, W, k, stride, pad, ld, dtype)   # recomputed gather
            dwp, pend = ops.linear_wgrad(dz, col, defer_reduce=True)                         # [Cout, ld] fp32
            unpack = ops.conv_weight_unpack_grad
        dw_box = [None]

        def emit_dw():
            """Unpack / scatter the reduced weight gradient into the parameter layout (and the sink).  Runs BEHIND the data
            gradient: the implicit weight gradient leaves its split-K reduce to that launch's grid tail (``pend``)."""
            if direct_dw:                                                    # the reduce wrote it in place
                if sw is not None:
                    sw.mark_written()
                else:
                    dw_box[0] = dw_master
            elif ctx.pair is not None:                                         # pixel-pair stem: adjoint of dvt_conv_weight_pairs
                kh_o, kw_o, pw_o, kwp = ctx.pair
                dw_pairs = unpack(dwp, w4)                                   # [Cout, 8, kh, kwp]
                if sw is not None and Cout == Cout_l:
                    ops.conv_weight_pairs_bwd(dw_pairs, Cout, Cin_l, kh_o, kw_o, pw_o, kwp, out=sw.buf, accumulate=not sw.fresh)
                    sw.mark_written()
                    dw_box[0] = None
                else:
                    dwo = ops.conv_weight_pairs_bwd(dw_pairs, Cout, Cin_l, kh_o, kw_o, pw_o, kwp)
                    if sw is not None:
                        ops.unpad3_f32(dwo, Cout_l, Cin_l, kh_o * kw_o, Cin_l, out=sw.buf, accumulate=not sw.fresh)
                        sw.mark_written()
                        dw_box[0] = None
                    else:
                        dw_box[0] = ops.unpad3_f32(dwo, Cout_l, Cin_l, kh_o * kw_o, Cin_l).view(wshape)
            elif padded:                                                     # full-width gradient, then the reference slice
                dw_full = unpack(dwp, w4)
                if sw is not None:
                    ops.unpad3_f32(dw_full, Cout_l, Cin_l, kh * kw, Cin, out=sw.buf, accumulate=not sw.fresh)
                    sw.mark_written()
                    dw_box[0] = None
                else:
                    dw_box[0] = ops.unpad3_f32(dw_full, Cout_l, Cin_l, kh * kw, Cin).view(wshape)
            elif sw is not None:
                unpack(dwp, w4, out=sw.buf, accumulate=not sw.fresh)
                sw.mark_written()
                dw_box[0] = None
            else:
                dw_box[0] = unpack(dwp, w4).view(wshape)

        dx = None
        (sh_, sw_), (ph_, pw_) = ops._pair(stride), ops._pair(pad)
        Ho, Wo = ops.conv_out_hw(H, W, k, stride, pad)
        # the shortcut's gradient (ctx.fork): joins the data gradient inside the kernel that writes it, where that kernel can
        joined = [dshort is None or not ctx.x_needs]
        if dshort is not None:
            dshort = _as(dshort.contiguous(), dz.dtype)

        def join_alias():
            """The full-size second gradient path for a kernel that adds it in its epilogue (None: nothing to add / compact)."""
            if joined[0] or ctx.fork != "alias":
                return None
            joined[0] = True
            return dshort.view(N * H * W, Cin)
        if (ctx.x_needs and ctx.implicit and ctx.geom0 is None and sh_ == 1 and sw_ == 1 and kh - 1 - ph_ >= 0
                and kw - 1 - pw_ >= 0
                and (Ho, Wo) == (H + 2 * ph_ - kh + 1, W + 2 * pw_ - kw + 1)):
            if ctx.w_ref is not None:                                    # [Cin, kh*kw*Cout], refreshed once per optimizer step
                wd = _packed_weight(ctx.w_ref, 1, Cout_l, Cin_l, kh, kw, Cout, Cin, 0, dtype)
            else:
                wd = ops.conv_weight_pack_dgrad(ctx.w4, dtype)
            pd = (kh - 1 - ph_, kw - 1 - pw_)
            if (HALO_CONV and Cin == 64 and Cout == 64 and (kh, kw) == (3, 3) and pd == (1, 1)
                    and ops.conv3x3_c64_supported(dz, wd, N, Ho, Wo)):
                dx = ops.conv3x3_c64(dz, wd, N, Ho, Wo, residual=join_alias())
            elif (HALO_CONV and (kh, kw) == (3, 3) and pd == (1, 1)
                    and ops.conv3x3_stream_supported(dz, wd, N, Ho, Wo, Cout, Cin)):
                dx = ops.conv3x3_stream(dz, wd, N, Ho, Wo, Cout, Cin, residual=join_alias())
            elif (HALO_CONV and (kh, kw) == (3, 1) and pd == (1, 0) and (joined[0] or ctx.fork != "alias")
                    and ops.conv3x1_stream_supported(dz, wd, N, Ho, Wo, Cout, Cin)):
                lk = ctx.in_affine[6] if (ctx.in_affine is not None and len(ctx.in_affine) > 6) else None
                if FUSED_MID_BN_BWD and lk is not None and lk.get("sinks") is not None and not ctx.in_affine[4]:
                    # the input is the spatial half's z under a virtual BatchNorm: its backward runs inside this data gradient
                    sgm, sbm = lk["sinks"]
                    if sgm is not None and sbm is not None and sgm.fresh == sbm.fresh:
                        dx, _, _ = ops.conv3x1_stream_bn_bwd(dz, wd, xc, ctx.in_affine[:6], N, Ho, Wo, lk["training"],
                                                             dgamma=sgm.buf.view(-1), dbeta=sbm.buf.view(-1),
                                                             accumulate=not sgm.fresh)
                        sgm.mark_written(); sbm.mark_written()
                        lk["grads"] = (None, None)
                    else:
                        dx, dgm, dbm = ops.conv3x1_stream_bn_bwd(dz, wd, xc, ctx.in_affine[:6], N, Ho, Wo, lk["training"])
                        if sgm is not None and sbm is not None:
                            _emit_into(sgm, dgm); _emit_into(sbm, dbm)
                            dgm = dbm = None
                        lk["grads"] = (dgm, dbm)
                    lk["done"] = True
                else:
                    dx = ops.conv3x1_stream(dz, wd, N, Ho, Wo, Cout, Cin)  # temporal half of layer 1's Conv2Plus1D: 64 -> 144
            elif (WINDOW_FWD and HALO_CONV and (kh, kw) == (3, 1) and pd == (1, 0) and Cin == 64 and Cout == 64
                    and (joined[0] or ctx.fork != "alias") and ops.conv3x1_fwd_supported(dz, wd, N, Ho, Wo, Cout, Cin)):
                dx = ops.conv3x1_fwd(dz, wd, N, Ho, Wo)           # the stem's temporal half: the window kernel on the data-gradient pack
            elif ops.conv2d_implicit_supported(dz, wd, N, Cout, Ho, Wo, Cin, k, 1, pd):
                dx = ops.conv2d_implicit(dz, wd, N, Cout, Ho, Wo, Cin, k, 1, pd, carry=pend,
                                         residual=join_alias())   # [N*H*W, Cin], no dcol / col2im
        if (ctx.x_needs and dx is None and STRIDED_IMPLICIT and ctx.implicit and ctx.geom0 is None and not nchw
                and (sh_, sw_) != (1, 1) and ctx.w_ref is not None and STRIDED_IMPLICIT_MIN_PIXELS <= N * H * W < (1 << 31)):
            # strided layer (custom_resnet.py:19-22 with stride 2; R(2+1)D's strided halves): the input pixels fall into
            # sh x sw parity classes, each the stride-1 convolution of dz with its own taps, scattered into dx by a row table
            classes = ops.strided_dgrad_classes(k, stride, pad, H, W)
            if classes is not None and all(ops.conv2d_implicit_k(Cout, Cin, nt) == nt[0] * nt[1] * Cout
                                           for (_, _, nt, _, _, _) in classes):
                full = compact = None
                if not joined[0]:                # the shortcut's gradient joins in the residual epilogue of the class launches
                    if ctx.fork == "alias":
                        full = dshort.view(N * H * W, Cin)
                    elif (int(ctx.fork), int(ctx.fork)) == (sh_, sw_):
                        compact = dshort.view(-1, Cin)      # a strided 1 x 1 shortcut touches class (0, 0) only
                    joined[0] = full is not None or compact is not None
                dx = torch.empty((N * H * W, Cin), dtype=dz.dtype, device=dz.device)
                order = sorted(classes, key=lambda c: c[2][0] * c[2][1])       # the class with the most taps carries the reduce
                for i, (a, b, nt, pq, (rh, rw), hq) in enumerate(order):
                    wc = _packed_class_weight(ctx.w_ref, Cout_l, Cin_l, kh, kw, Cout, Cin, (sh_, sw_, rh, rw), dtype)
                    res, rc = (full, False) if full is not None else ((compact, True) if (compact is not None and a == 0 and b == 0)
                                                                      else (None, False))
                    ops.conv2d_implicit(dz, wc, N, Cout, Ho, Wo, Cin, nt, 1, pq, out=dx, out_hw=hq,
                                        out_rows=ops.strided_class_rows(N, H, W, sh_, sw_, a, b, dz.device), residual=res,
                                        residual_compact=rc, carry=pend if i == len(order) - 1 else None)
        if ctx.x_needs and dx is None and (nchw or ctx.geom0 is not None) and ctx.dx_frames:
            if ctx.geom0 is not None:            # implicit stem: the frames' own geometry and the plain [Cout, taps * Cin] pack
                N0, C0, H0, W0, k0, s0, p0 = ctx.geom0
                (kh0, kw0) = ops._pair(k0)
                wq = _packed_weight(ctx.w_stem, 0, Cout_l, Cin_l, kh0, kw0, Cout, C0, _kpad(kh0 * kw0 * C0, dtype), dtype)
            else:
                (N0, C0, H0, W0, k0, s0, p0), wq = (N, Cin, H, W, k, stride, pad), wp
            Ho0, Wo0 = ops.conv_out_hw(H0, W0, k0, s0, p0)
            dx = ops.zeros(ctx.x_shape, ctx.x_dtype, dz.device).view(N0, C0, H0, W0)
            hw = Ho0 * Wo0
            for f0, cnt in ctx.dx_frames:                                # only the frames whose gradient is read
                dcol = ops.linear_dgrad(dz[f0 * hw:(f0 + cnt) * hw], wq)
                ops.copy_(dx[f0:f0 + cnt], ops.col2im_nchw(dcol, cnt, C0, H0, W0, k0, s0, p0, ctx.x_dtype))
            dx = dx.view(ctx.x_shape)
        if ctx.x_needs and dx is None:
            dcol = ops.linear_dgrad(dz, wp, carry=pend)                  # [rows, ld]
            if direct:
                dx = dcol
            elif nchw:      # gradient w.r.t. the raw NCHW frames (pixel-space CLS clip, frame_transformer.py:105)
                dx = ops.col2im_nchw(dcol, N, Cin, H, W, k, stride, pad, ctx.x_dtype).view(ctx.x_shape)
            elif not joined[0] and Cin % 8 == 0 and ld % 8 == 0:          # the shortcut's gradient joins inside col2im
                joined[0] = True
                if ctx.fork == "alias":
                    dx = ops.col2im(dcol, N, Cin, H, W, k, stride, pad, add=dshort.view(N * H * W, Cin))
                else:
                    dx = ops.col2im(dcol, N, Cin, H, W, k, stride, pad, add=dshort, add_stride=int(ctx.fork))
            else:
                dx = ops.col2im(dcol, N, Cin, H, W, k, stride, pad)
        if not joined[0]:                        # a path without a fused form (1x1 / stride-1 layers, the fp32 parity mode)
            dfull = dshort.view(dx.shape) if ctx.fork == "alias" else ops.col2im(dshort, N, Cin, H, W, 1, int(ctx.fork), 0)
            dx = ops.add(dx.contiguous(), dfull.view(dx.shape))
        ops.splitk_reduce_pending(pend)          # nobody carried it (no data gradient wanted, or the halo kernel computed it)
        emit_dw()
        dw = dw_box[0]
        return dx, dw, dgam, dbet, dres, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None


def conv_bn_act(x, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d, geom, *, relu: bool, residual=None,
                dtype=torch.bfloat16, pool: bool = False, fork=None, stride=None):
    """geom = (N, Cin, H, W, nchw).  Kernel size / stride / padding come from ``conv``.  pool: the layer is followed by
    MaxPool2d(3, 2, 1); the pooled map is returned (see _ConvBnAct)."""
    N, Cin, H, W, nchw = geom
    assert conv.bias is None and conv.groups == 1
    return conv_bn_act_raw(x, conv.weight, bn, geom, tuple(conv.kernel_size), tuple(conv.stride) if stride is None else stride,
                           tuple(conv.padding), relu=relu, residual=residual, dtype=dtype, pool=pool, fork=fork)


def conv_bn_act_raw(x, weight, bn, geom, k, stride, pad, *, relu: bool, residual=None, dtype=torch.bfloat16,
                    cpad: int = 0, dx_frames=None, pool: bool = False, fork=None, in_affine=None, defer_apply: bool = False):
    """Same with an explicit 2-D kernel geometry (k, stride, pad: ints or (h, w) pairs); ``weight`` may be a
    Conv3d weight whose singleton kernel axis is dropped by the caller's choice of ``k``
    (factorised R(2+1)D convolutions).  ``bn``: BatchNorm2d/3d parameter container.
    defer_apply: no BatchNorm-apply pass -- returns (z, affine) with affine = (mean, invstd, gamma, beta, c_valid, relu) for the
    ``in_affine`` of the layer behind, whose kernels then form the normalised activation in LDS (see _ConvBnAct)."""
    N, Cin, H, W, nchw = geom
    training = bn.training or bn.running_mean is None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    if defer_apply:
        joins = fork is not None and torch.is_grad_enabled() and x.requires_grad
        link = {}
        outs = _ConvBnAct.apply(x, weight, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                                (N, Cin, H, W, k, stride, pad, nchw), relu, training, momentum, bn.eps, dtype, cpad,
                                dx_frames, pool, fork if joins else None, in_affine, True, link)
        z, (mean, invstd) = outs[0], outs[-2:]
        cout_l = weight.shape[0]
        affine = (mean, invstd, _f32(bn.weight), _f32(bn.bias), cout_l if z.shape[1] != cout_l else 0, relu, link)
        if fork is None:
            return z, affine
        second = outs[1] if joins else (x if fork == "alias" else subsample_nhwc(x, N, Cin, H, W, int(fork)))
        return (z, second), affine
    if fork is not None and not (torch.is_grad_enabled() and x.requires_grad):       # nothing to join: hand the shortcut its input
        y = _ConvBnAct.apply(x, weight, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                             (N, Cin, H, W, k, stride, pad, nchw), relu, training, momentum, bn.eps, dtype, cpad, dx_frames, pool,
                             None, in_affine)
        return y, (x if fork == "alias" else subsample_nhwc(x, N, Cin, H, W, int(fork)))
    return _ConvBnAct.apply(x, weight, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                            (N, Cin, H, W, k, stride, pad, nchw), relu, training, momentum, bn.eps, dtype, cpad, dx_frames, pool,
                            fork, in_affine)


class _Subsample(torch.autograd.Function):
    """Strided spatial subsampling of an NHWC matrix (the gather of a strided 1x1 convolution)."""

    @staticmethod
    def forward(ctx, x, geom):
        N, Cc, H, W, stride = geom
        ctx.geom = geom
        return ops.im2col(x.contiguous(), False, N, Cc, H, W, 1, stride, 0, Cc, x.dtype)

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W, stride = ctx.geom
        return ops.col2im(dy.contiguous(), N, Cc, H, W, 1, stride, 0), None


def subsample_nhwc(x, N, Cc, H, W, stride):
    """stride: int or (h, w).  Returns [N*Ho*Wo, C]."""
    return _Subsample.apply(x, (N, Cc, H, W, stride))


class _Im2Col(torch.autograd.Function):
    """NHWC matrix [N*H*W, C] -> column matrix [N*Ho*Wo, kh*kw*C] (adjoint: col2im)."""

    @staticmethod
    def forward(ctx, x, geom):
        N, Cc, H, W, k, stride, pad = geom
        ctx.geom = geom
        (kh, kw) = ops._pair(k)
        return ops.im2col(x.contiguous(), False, N, Cc, H, W, k, stride, pad, kh * kw * Cc, x.dtype)

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W, k, stride, pad = ctx.geom
        return ops.col2im(dy.contiguous(), N, Cc, H, W, k, stride, pad), None


def im2col_nhwc(x, N, Cc, H, W, k, stride, pad):
    return _Im2Col.apply(x, (N, Cc, H, W, k, stride, pad))


class _Col2Im(torch.autograd.Function):
    """Column matrix [N*Ho*Wo, kh*kw*C] -> NHWC matrix [N*H*W, C] (overlap-add; with k == stride this is the
    pixel shuffle of a transposed convolution).  Adjoint: im2col."""

    @staticmethod
    def forward(ctx, col, geom):
        N, Cc, H, W, k, stride, pad = geom
        ctx.geom = geom
        return ops.col2im(col.contiguous(), N, Cc, H, W, k, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W, k, stride, pad = ctx.geom
        (kh, kw) = ops._pair(k)
        return ops.im2col(dy.contiguous(), False, N, Cc, H, W, k, stride, pad, kh * kw * Cc, dy.dtype), None


def col2im_nhwc(col, N, Cc, H, W, k, stride, pad):
    return _Col2Im.apply(col, (N, Cc, H, W, k, stride, pad))


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geom):
        N, Cc, H, W, k, stride, pad = geom
        y, idx = ops.maxpool_fwd(x.contiguous(), N, Cc, H, W, k, stride, pad)
        ctx.save_for_backward(idx)
        ctx.geom = geom
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, Cc, H, W, k, stride, pad = ctx.geom
        return ops.maxpool_bwd(dy.contiguous(), idx, N, Cc, H, W, k, stride, pad), None


def maxpool_nhwc(x, N, Cc, H, W, k, stride, pad):
    return _MaxPool.apply(x, (N, Cc, H, W, k, stride, pad))


class _Transpose12(torch.autograd.Function):
    """[B, R, C] <-> [B, C, R] (NHWC <-> NCHW at the module boundary)."""

    @staticmethod
    def forward(ctx, x):
        return ops.transpose_last2(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.transpose_last2(dy)


def transpose_last2(x):
    return _Transpose12.apply(x)


# ---------------------------------------------------------------------------
# activation checkpointing (BASELINE configs[4]: T=64, 288^2 long-clip stress)
# ---------------------------------------------------------------------------
class _Checkpoint(torch.autograd.Function):
    """Runs ``fn(x)`` without recording, keeps only ``x``; backward re-runs ``fn`` with recording and
    back-propagates through it.  Parameter gradients are produced by the inner backward (into the
    ``dp.FlatParameters`` sinks, or accumulated into ``.grad`` by autograd) -- the ``params`` inputs only
    make the output require grad when ``x`` itself does not."""

    @staticmethod
    def forward(ctx, fn, x, *params):
        ctx.fn = fn
        ctx.save_for_backward(x)
        ctx.site = _rng.site                 # the recomputation must draw the same dropout masks
        with torch.no_grad():
            return fn(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        xin = x.detach().requires_grad_(True)
        site_now, _rng.site = _rng.site, ctx.site
        with torch.enable_grad():
            y = ctx.fn(xin)
        _rng.site = site_now
        torch.autograd.backward(y, dy)
        return (None, xin.grad) + (None,) * (len(ctx.needs_input_grad) - 2)


def checkpoint(fn, x: Tensor, params=()) -> Tensor:
    """y = fn(x) storing only x for backward (one extra forward of ``fn`` per step)."""
    if not torch.is_grad_enabled():
        return fn(x)
    return _Checkpoint.apply(fn, x, *params)


# ---------------------------------------------------------------------------
# Multi-modal gating and the contrastive objective (collabgating.py, losses/ntxent.py)
# ---------------------------------------------------------------------------
class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        y, inv = ops.l2norm_rows_fwd(x.reshape(-1, x.shape[-1]), eps)
        ctx.save_for_backward(y, inv)
        ctx.eps, ctx.shape = eps, tuple(x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.l2norm_rows_bwd(_as(dy.reshape(y.shape), y.dtype), y, inv, ctx.eps).view(ctx.shape), None


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    """F.normalize(x, dim=-1) (collabgating.py:70)."""
    return _L2Normalize.apply(x, eps)


class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.gate_fwd(a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        return ops.gate_bwd(_as(dy, a.dtype), a, b)


def gate(a: Tensor, b: Tensor) -> Tensor:
    """a * sigmoid(b): ``F.glu(cat(x, x + x1), -1)`` with a = x, b = x + x1 (collabgating.py:83-86)."""
    return _Gate.apply(a, b)


class _Contrastive(torch.autograd.Function):
    """ContrastiveLoss.forward (ntxent.py:53-75) on reps = cat(z_i, z_j) [M, D] (fp32 arithmetic)."""

    @staticmethod
    def forward(ctx, reps, temperature):
        r32 = _as(reps.contiguous(), torch.float32)
        zn, inv = ops.l2norm_rows_fwd(r32, 1e-8)                         # F.cosine_similarity eps
        M, D = zn.shape
        sim = ops.gemm(zn, zn, M, M, D, a_kmajor=True, b_kmajor=True, lda=D, ldb=D, out_dtype=torch.float32)
        loss, lse = ops.contrastive_fwd(sim, temperature)
        ctx.save_for_backward(zn, inv, sim, lse)
        ctx.t, ctx.dtype = temperature, reps.dtype
        return loss

    @staticmethod
    def backward(ctx, gloss):
        zn, inv, sim, lse = ctx.saved_tensors
        M, D = zn.shape
        g = _as(gloss.contiguous().reshape(1), torch.float32)
        dsim = ops.contrastive_bwd(sim, lse, ctx.t, g)
        # sim = zn zn^T  =>  dzn = dsim zn + dsim^T zn
        dzn = ops.gemm(dsim, zn, M, D, M, a_kmajor=True, b_kmajor=False, lda=M, ldb=D, out_dtype=torch.float32)
        ops.gemm(dsim, zn, M, D, M, a_kmajor=False, b_kmajor=False, lda=M, ldb=D, out=dzn, out_dtype=torch.float32,
                 accumulate=True)
        return _as(ops.l2norm_rows_bwd(dzn, zn, inv, 1e-8), ctx.dtype), None


def contrastive_loss(z_i: Tensor, z_j: Tensor, temperature: float) -> Tensor:
    return _Contrastive.apply(concat_rows(z_i, z_j), temperature)


# ---------------------------------------------------------------------------
# Dropout (training mode)
# ---------------------------------------------------------------------------
class DropoutRng:
    """Device-resident generator state {seed, step base offset} plus the host-side offset of the next dropout site
    of the current step.  ``F.manual_seed`` / ``F.next_step`` are the only controls: call ``next_step()`` once per
    optimisation step (``dp.FlatParameters`` optimizer steps do) -- it advances the base on the device, so the masks
    change from step to step even when the step is a replayed hipGraph."""

    def __init__(self):
        self.state = None
        self.site = 0
        self.seed = 1130                                   # src/main.py:25

    def tensor(self, device) -> Tensor:
        if self.state is None or self.state.device != device:
            self.state = torch.tensor([self.seed, 0], dtype=torch.int64, device=device)
        return self.state

    def take(self, n: int) -> int:
        off = self.site
        self.site += (n + 3) // 4
        return off


_rng = DropoutRng()


def manual_seed(seed: int) -> None:
    _rng.seed, _rng.state, _rng.site = int(seed), None, 0


def next_step() -> None:
    """Advance the dropout generator past every site drawn in this step (device-side add) and restart site numbering."""
    if _rng.state is not None and _rng.site:
        ops.rng_advance_(_rng.state, _rng.site)
    _rng.site = 0


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        st = _rng.tensor(x.device)
        ctx.p, ctx.off, ctx.state = p, _rng.take(x.numel()), st
        return ops.dropout(x, p, st, ctx.off)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.state, ctx.off), None


class _DropoutAdd(torch.autograd.Function):
    """res + dropout(x): one launch (the mask of ``_Dropout`` at the same RNG site)."""

    @staticmethod
    def forward(ctx, x, res, p):
        st = _rng.tensor(x.device)
        ctx.p, ctx.off, ctx.state = p, _rng.take(x.numel()), st
        return ops.dropout_fused(x, p, st, ctx.off, residual=res.contiguous())

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return ops.dropout(dy, ctx.p, ctx.state, ctx.off), dy, None


class _ReluDropout(torch.autograd.Function):
    """dropout(relu(x)): one launch forward, one backward (gated by the saved output: no mask is re-drawn)."""

    @staticmethod
    def forward(ctx, x, p):
        st = _rng.tensor(x.device)
        y = ops.dropout_fused(x, p, st, _rng.take(x.numel()), relu=True)
        ctx.p = p
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.dropout_fused(dy.contiguous(), ctx.p, None, 0, gate=y), None


def dropout_add(x: Tensor, res: Tensor, p: float, training: bool) -> Tensor:
    """res + nn.Dropout(p)(x) (the residual connections of nn.TransformerEncoderLayer in training mode)."""
    if not training or p <= 0.0:
        return add(res, x)
    if p >= 1.0:
        raise ValueError("dropout p must be < 1")
    return _DropoutAdd.apply(x, res, float(p))


def relu_dropout(x: Tensor, p: float, training: bool) -> Tensor:
    """nn.Dropout(p)(relu(x)) (the feed-forward hidden state of nn.TransformerEncoderLayer in training mode)."""
    if not training or p <= 0.0:
        return relu(x)
    if p >= 1.0:
        raise ValueError("dropout p must be < 1")
    return _ReluDropout.apply(x, float(p))


def dropout(x: Tensor, p: float, training: bool) -> Tensor:
    """nn.Dropout(p)(x): identity unless ``training`` and p > 0."""
    if not training or p <= 0.0:
        return x
    if p >= 1.0:
        raise ValueError("dropout p must be < 1")
    return _Dropout.apply(x, float(p))
