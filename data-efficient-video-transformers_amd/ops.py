"""Raw (autograd-free) Python wrappers over the C ABI.

Every function launches hand-written HIP kernels from libdvt_hip.so on the
current torch stream; torch is used only to own device memory.  Nothing here
falls back to torch arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}
_TORCH_DT = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.F16: torch.float16}


def dt(t: Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"dtype {t.dtype} is not supported by libdvt_hip (float32 / bfloat16)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts: Optional[Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libdvt_hip operators need tensors on the GPU (HIP device); "
                               "there is no CPU path")


_ws = {}

# Optional live profiler (bench.py): an object with begin(key, flops) / end(key) that
# records HIP events on the current stream around selected launches.
_profiler = None


def set_profiler(p) -> None:
    global _profiler
    _profiler = p


class _timed:
    """``with _timed(key, units):`` -- bracket a launch with the profiler's events when one is installed
    (units = algorithmic bytes for the HBM-bound kernels, flops for GEMMs)."""

    __slots__ = ("key", "units", "prof")

    def __init__(self, key, units):
        self.key, self.units, self.prof = key, units, _profiler

    def __enter__(self):
        if self.prof is not None:
            self.prof.begin(self.key, self.units)

    def __exit__(self, *exc):
        if self.prof is not None:
            self.prof.end(self.key)
        return False



def workspace(nbytes: int, device, slot: str = "main") -> Optional[Tensor]:
    """Grow-only scratch buffer per (device, stream, slot).  Kernels on one stream run in
    order, so consecutive operators on it may share the buffer; another stream gets its own."""
    if nbytes <= 0:
        return None
    key = (str(device), slot, _stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


_live_deferred = {}      # (device, stream) -> the SplitKPending whose slabs occupy the "deferred" scratch slot


def _deferred_workspace(nbytes: int, device, pending) -> Optional[Tensor]:
    """Scratch slot for the slabs of a deferred split-K reduce.  One slot per (device, stream): a second deferral while an
    earlier pending reduce is still outstanding would overwrite that one's slabs -- refuse it instead of producing a
    silently wrong weight gradient (callers resolve a pending reduce -- ``carry=`` / ``splitk_reduce_pending`` -- before
    they defer the next)."""
    key = (str(device), _stream())
    prev = _live_deferred.get(key)
    if prev is not None and prev.valid:
        raise RuntimeError("a deferred split-K reduce is still pending on this stream: carry it (carry=) or call "
                           "ops.splitk_reduce_pending before deferring another one")
    _live_deferred[key] = pending
    return workspace(nbytes, device, slot="deferred")


# ------------------------------------------------------------------ elementwise
def cast(src: Tensor, dtype: torch.dtype) -> Tensor:
    _need_cuda(src)
    src = src.contiguous()
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    L.check(L.load().dvt_cast(src.data_ptr(), dt(src), out.data_ptr(), _DT[dtype], src.numel(), _stream()),
            "dvt_cast")
    return out


def zeros(shape, dtype: torch.dtype, device) -> Tensor:
    """A zero-filled tensor without an ATen fill kernel (hipMemsetAsync on the launch stream)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    if not t.is_cuda:
        raise RuntimeError("dvt_amd ops have no CPU path")
    L.check(L.load().dvt_zero(t.data_ptr(), t.numel() * t.element_size(), _stream()), "dvt_zero")
    return t


def add(a: Tensor, b: Tensor) -> Tensor:
    _need_cuda(a, b)
    assert a.shape == b.shape and a.dtype == b.dtype
    a = a.contiguous(); b = b.contiguous()
    out = torch.empty_like(a)
    L.check(L.load().dvt_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), dt(a), _stream()), "dvt_add")
    return out


def add_rowtable(x: Tensor, table: Tensor, rows_per_entry: int) -> Tensor:
    """out[r] = x[r] + table[r // rows_per_entry]; x [rows, d], table [rows/rows_per_entry, d] f32."""
    _need_cuda(x, table)
    assert x.is_contiguous() and table.is_contiguous() and table.dtype == torch.float32
    d = x.shape[-1]
    rows = x.numel() // d
    assert table.shape[-1] == d and table.numel() // d * rows_per_entry >= rows
    out = torch.empty_like(x)
    L.check(L.load().dvt_add_rowtable(x.data_ptr(), table.data_ptr(), out.data_ptr(), rows, d, rows_per_entry,
                                      dt(x), _stream()), "dvt_add_rowtable")
    return out


def copy_(dst: Tensor, src: Tensor) -> Tensor:
    """Contiguous device copy (optionally converting dtype) through dvt_cast."""
    _need_cuda(dst, src)
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
    L.check(L.load().dvt_cast(src.data_ptr(), dt(src), dst.data_ptr(), dt(dst), src.numel(), _stream()), "dvt_cast")
    return dst


def copy2d(src: Tensor, dst: Tensor, rows: int, cols: int, src_ld: int, dst_ld: int) -> None:
    """dst[r*dst_ld + c] = src[r*src_ld + c] on the raw storages starting at the tensors' data pointers."""
    _need_cuda(src, dst)
    assert src.dtype == dst.dtype
    L.check(L.load().dvt_copy2d(src.data_ptr(), dst.data_ptr(), rows, cols, src_ld, dst_ld, dt(src), _stream()),
            "dvt_copy2d")


def rows_sum(src: Tensor, row_stride: int, rows: int, cols: int, *, out: Optional[Tensor] = None,
             accumulate: bool = False) -> Tensor:
    _need_cuda(src)
    if out is None:
        assert not accumulate
        out = torch.empty((cols,), dtype=torch.float32, device=src.device)
    L.check(L.load().dvt_rows_sum(src.data_ptr(), row_stride, rows, cols, out.data_ptr(), dt(src), int(accumulate),
                                  _stream()), "dvt_rows_sum")
    return out


def permute_021(x: Tensor) -> Tensor:
    """[A, B, C] -> [B, A, C] (contiguous)."""
    _need_cuda(x)
    x = x.contiguous()
    A, B, Cc = x.shape
    out = torch.empty((B, A, Cc), dtype=x.dtype, device=x.device)
    L.check(L.load().dvt_permute_021(x.data_ptr(), out.data_ptr(), A, B, Cc, dt(x), _stream()), "dvt_permute_021")
    return out


def axpby_f32_(dst: Tensor, src: Tensor, alpha: float = 1.0, beta: float = 1.0) -> Tensor:
    """dst(f32) = beta*dst + alpha*src."""
    _need_cuda(dst, src)
    assert dst.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous()
    assert dst.numel() == src.numel()
    L.check(L.load().dvt_axpby_f32(src.data_ptr(), dt(src), alpha, dst.data_ptr(), beta, dst.numel(), _stream()),
            "dvt_axpby_f32")
    return dst


ACT_GELU, ACT_RELU, ACT_SIGMOID = 1, 2, 3


def act_fwd(x: Tensor, act: int) -> Tensor:
    _need_cuda(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.load().dvt_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), act, dt(x), _stream()), "dvt_act_fwd")
    return y


def act_bwd(dy: Tensor, x: Tensor, act: int) -> Tensor:
    _need_cuda(dy, x)
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    L.check(L.load().dvt_act_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), act, dt(x), _stream()),
            "dvt_act_bwd")
    return dx


def patchify(x: Tensor, patch: int, out_dtype: torch.dtype) -> Tensor:
    """[..., C, H, W] -> [frames * n, P*P*C]  (vit.py:90 ordering)."""
    _need_cuda(x)
    x = x.contiguous()
    Cc, H, W = x.shape[-3:]
    frames = x.numel() // (Cc * H * W)
    n = (H // patch) * (W // patch)
    out = torch.empty((frames * n, patch * patch * Cc), dtype=out_dtype, device=x.device)
    with _timed(("hbm", "patchify", x.numel()), x.numel() * x.element_size() + out.numel() * out.element_size()):
        L.check(L.load().dvt_patchify(x.data_ptr(), dt(x), out.data_ptr(), _DT[out_dtype], frames, Cc, H, W, patch,
                                      _stream()), "dvt_patchify")
    return out


def patchify_bwd(dout: Tensor, shape, patch: int, dx_dtype: torch.dtype) -> Tensor:
    _need_cuda(dout)
    dout = dout.contiguous()
    Cc, H, W = shape[-3:]
    frames = 1
    for s in shape[:-3]:
        frames *= s
    dx = torch.empty(tuple(shape), dtype=dx_dtype, device=dout.device)
    L.check(L.load().dvt_patchify_bwd(dout.data_ptr(), dt(dout), dx.data_ptr(), _DT[dx_dtype], frames, Cc, H, W,
                                      patch, _stream()), "dvt_patchify_bwd")
    return dx


def tokens_assemble_fwd(emb: Tensor, cls: Tensor, pos: Tensor, S: int, T: int, n: int) -> Tensor:
    """emb [S*n, d]; cls [d] f32; pos [T, rows>=n+1, d] f32 -> [S, n+1, d]."""
    _need_cuda(emb, cls, pos)
    d = emb.shape[-1]
    assert emb.is_contiguous() and cls.is_contiguous() and pos.is_contiguous()
    assert cls.dtype == torch.float32 and pos.dtype == torch.float32
    out = torch.empty((S, n + 1, d), dtype=emb.dtype, device=emb.device)
    with _timed(("hbm", "tokens_assemble_fwd", emb.numel()), (emb.numel() + out.numel()) * emb.element_size() + pos.numel() * 4):
        L.check(L.load().dvt_tokens_assemble_fwd(emb.data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), S, T,
                                                 n, d, pos.shape[-2], dt(emb), _stream()), "dvt_tokens_assemble_fwd")
    return out


def tokens_assemble_bwd(dout: Tensor, T: int, pos_rows: int, *, dcls: Optional[Tensor] = None,
                        dpos: Optional[Tensor] = None, accumulate: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """dcls / dpos may be caller-owned fp32 destinations (gradient sinks)."""
    _need_cuda(dout)
    dout = dout.contiguous()
    S, n1, d = dout.shape
    n = n1 - 1
    demb = torch.empty((S * n, d), dtype=dout.dtype, device=dout.device)
    if dcls is None:
        assert not accumulate
        dcls = torch.empty((d,), dtype=torch.float32, device=dout.device)
    if dpos is None:
        assert not accumulate
        dpos = torch.empty((T, pos_rows, d), dtype=torch.float32, device=dout.device)
    assert dcls.is_contiguous() and dpos.is_contiguous() and dcls.numel() == d and dpos.numel() == T * pos_rows * d
    L.check(L.load().dvt_tokens_assemble_bwd(dout.data_ptr(), demb.data_ptr(), dcls.data_ptr(), dpos.data_ptr(), S,
                                             T, n, d, pos_rows, dt(dout), int(accumulate), _stream()),
            "dvt_tokens_assemble_bwd")
    return demb, dcls, dpos


def rows_gather_fwd(src: Tensor, row_stride: int, tok: Optional[Tensor], B: int, T: int, d: int) -> Tensor:
    _need_cuda(src, tok)
    lead = 0 if tok is None else 1
    out = torch.empty((B, T + lead, d), dtype=src.dtype, device=src.device)
    L.check(L.load().dvt_rows_gather_fwd(src.data_ptr(), row_stride, _p(tok), out.data_ptr(), B, T, d, dt(src),
                                         _stream()), "dvt_rows_gather_fwd")
    return out


def rows_gather_bwd(dout: Tensor, dsrc: Tensor, row_stride: int, want_tok: bool, B: int, T: int,
                    d: int, *, dtok: Optional[Tensor] = None, accumulate: bool = False) -> Optional[Tensor]:
    """Scatters dout rows into ``dsrc`` (pre-zeroed by the caller where needed)."""
    _need_cuda(dout, dsrc)
    dout = dout.contiguous()
    if want_tok and dtok is None:
        assert not accumulate
        dtok = torch.empty((d,), dtype=torch.float32, device=dout.device)
    if not want_tok:
        dtok = None
    L.check(L.load().dvt_rows_gather_bwd(dout.data_ptr(), dsrc.data_ptr(), row_stride, _p(dtok), B, T, d, dt(dout),
                                         int(accumulate), _stream()), "dvt_rows_gather_bwd")
    return dtok


def mean_rows_fwd(x: Tensor, scale: Optional[float] = None) -> Tensor:
    """[B, L, d] -> [B, d]: scale * sum over L (default scale 1/L = mean)."""
    _need_cuda(x)
    x = x.contiguous()
    B, Ln, d = x.shape
    out = torch.empty((B, d), dtype=x.dtype, device=x.device)
    sc = 1.0 / Ln if scale is None else scale
    L.check(L.load().dvt_mean_rows_fwd(x.data_ptr(), out.data_ptr(), B, Ln, d, sc, dt(x), _stream()),
            "dvt_mean_rows_fwd")
    return out


def mean_rows_bwd(dout: Tensor, Ln: int, scale: Optional[float] = None) -> Tensor:
    _need_cuda(dout)
    dout = dout.contiguous()
    B, d = dout.shape
    dx = torch.empty((B, Ln, d), dtype=dout.dtype, device=dout.device)
    sc = 1.0 / Ln if scale is None else scale
    L.check(L.load().dvt_mean_rows_bwd(dout.data_ptr(), dx.data_ptr(), B, Ln, d, sc, dt(dout), _stream()),
            "dvt_mean_rows_bwd")
    return dx


# ------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5, *, rows=None,
                  out: Optional[Tensor] = None, out_rows=None, out_dtype: Optional[torch.dtype] = None
                  ) -> Tuple[Tensor, Tensor, Tensor]:
    """Dense (rows is None): x [..., d] contiguous -> y like x.
    Strided: rows = (n0, n1, xs0, xs1) selects row (i0,i1) of ``x`` at element
    i0*xs0 + i1*xs1; the output row goes to ``out`` at i0*ys0 + i1*ys1 with
    out_rows = (ys0, ys1) (default: a dense [n0*n1, d] tensor).
    out_dtype: element type of y when it differs from x's (fp32 stream <-> 16-bit GEMM operand, see dvt_layernorm_fwd_mixed)."""
    _need_cuda(x, gamma, beta)
    d = x.shape[-1]
    ydt = x.dtype if out_dtype is None else out_dtype
    if rows is None:
        assert x.is_contiguous()
        n0, n1, xs0, xs1 = x.numel() // d, 1, d, 0
        if out is None:
            out = torch.empty(x.shape, dtype=ydt, device=x.device)
        ys0, ys1 = d, 0
    else:
        n0, n1, xs0, xs1 = rows
        if out is None:
            out = torch.empty((n0 * n1, d), dtype=ydt, device=x.device)
        ys0, ys1 = out_rows if out_rows is not None else (n1 * d, d)
    assert out.dtype == ydt
    nrows = n0 * n1
    mean = torch.empty((nrows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((nrows,), dtype=torch.float32, device=x.device)
    with _timed(("hbm", "layernorm_fwd", nrows), nrows * (d * (x.element_size() + out.element_size()) + 8)):
        L.check(L.load().dvt_layernorm_fwd_mixed(x.data_ptr(), dt(x), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                                 dt(out), mean.data_ptr(), rstd.data_ptr(), n0, n1, d, xs0, xs1, ys0, ys1,
                                                 eps, _stream()), "dvt_layernorm_fwd")
    return out, mean, rstd


# dgamma / dbeta reduces left undone by layernorm_bwd(defer=...): (pending descriptor, callback) pairs, performed by ONE
# launch in layernorm_flush() -- the launch-bound zone of the step has a dozen LayerNorms whose reduces were a launch each
_ln_deferred = []
LN_DEFER_MAX = 24


def layernorm_flush() -> None:
    """Perform every deferred dgamma / dbeta reduce (one launch per 32) and run their completion callbacks."""
    global _ln_deferred
    if not _ln_deferred:
        return
    todo, _ln_deferred = _ln_deferred, []
    arr = (L.LnPending * len(todo))(*[pn for pn, _ in todo])
    L.check(L.load().dvt_layernorm_reduce_group(C.cast(arr, C.c_void_p), len(todo), _stream()), "dvt_layernorm_reduce_group")
    for pn, done in todo:
        pn.valid = 0
        if done is not None:
            done()


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, *,
                  dx_add: Optional[Tensor] = None, rows=None, dy_rows=None,
                  dx: Optional[Tensor] = None, dg: Optional[Tensor] = None, db: Optional[Tensor] = None,
                  accumulate: bool = False, accumulate_beta: Optional[bool] = None,
                  dy_first: Optional[Tensor] = None, dx_first: Optional[Tensor] = None,
                  dx_dtype: Optional[torch.dtype] = None, dx_lp: Optional[torch.dtype] = None, defer=None):
    """``accumulate`` applies to dgamma (and to dbeta unless ``accumulate_beta`` is given).
    ``dy_first`` / ``dx_first`` [n0, d] (row stride free): added to dy / dx of row (i0, 0) only.
    Element types: dy's, x's and dx's (``dx_dtype``, default x's) may differ as dvt_layernorm_bwd_ex allows; ``dx_lp``: also
    return a copy of dx in that 16-bit type (-> (dx, dg, db, dx_copy)).  ``defer``: a callable (or True): the dgamma / dbeta
    reduce is left to ``layernorm_flush()`` (one launch for many layers), which calls it when dg / db are final."""
    _need_cuda(dy, x, gamma, mean, rstd, dx_add, dy_first, dx_first)
    d = x.shape[-1]
    if rows is None:
        assert x.is_contiguous() and dy.is_contiguous()
        n0, n1, xs0, xs1 = x.numel() // d, 1, d, 0
        ys0, ys1 = d, 0
    else:
        n0, n1, xs0, xs1 = rows
        ys0, ys1 = dy_rows if dy_rows is not None else (d * n1, d)
    dxt = x.dtype if dx_dtype is None else dx_dtype
    assert dy_first is None or (dy_first.dim() == 2 and dy_first.shape == (n0, d) and dy_first.stride(1) == 1 and dy_first.dtype == dy.dtype)
    assert dx_first is None or (dx_first.dim() == 2 and dx_first.shape == (n0, d) and dx_first.stride(1) == 1 and dx_first.dtype == dxt)
    if dx is None:
        dx = torch.empty(x.shape, dtype=dxt, device=x.device)
    assert dx.dtype == dxt and (dx_add is None or dx_add.dtype == dxt)
    if dg is None or db is None:
        assert not accumulate and not accumulate_beta and dg is None and db is None
        dg = torch.empty((d,), dtype=torch.float32, device=x.device)
        db = torch.empty((d,), dtype=torch.float32, device=x.device)
    acc_b = accumulate if accumulate_beta is None else accumulate_beta
    lib = L.load()
    q = L.LnBwdDesc()
    q.dy, q.dy_dtype, q.x, q.x_dtype = dy.data_ptr(), dt(dy), x.data_ptr(), dt(x)
    q.gamma, q.mean, q.rstd = gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    q.dx_add, q.dx, q.dx_dtype = _p(dx_add), dx.data_ptr(), _DT[dxt]
    lp = None
    if dx_lp is not None:
        lp = torch.empty(x.shape, dtype=dx_lp, device=x.device)
        q.dx_lp, q.dx_lp_dtype = lp.data_ptr(), _DT[dx_lp]
    q.dgamma, q.dbeta = dg.data_ptr(), db.data_ptr()
    q.n0, q.n1, q.d, q.xs0, q.xs1, q.ys0, q.ys1 = n0, n1, d, xs0, xs1, ys0, ys1
    q.dy_first, q.dy_first_stride = _p(dy_first), dy_first.stride(0) if dy_first is not None else 0
    q.dx_first, q.dx_first_stride = _p(dx_first), dx_first.stride(0) if dx_first is not None else 0
    q.accumulate_gamma, q.accumulate_beta = int(accumulate), int(acc_b)
    pending = None
    if defer:
        # the partial rows must outlive this call: a buffer of their own, kept alive by the pending record
        ws = torch.empty((lib.dvt_layernorm_bwd_partial_bytes(n0 * n1, d),), dtype=torch.uint8, device=x.device)
        pending = L.LnPending()
        pending._keep = (ws, dg, db)
        q.defer_reduce, q.pending = 1, C.pointer(pending)
    else:
        ws = workspace(lib.dvt_layernorm_bwd_workspace_bytes(d), x.device)
    q.workspace = ws.data_ptr()
    nb = n0 * n1 * (d * (dy.element_size() + x.element_size() + dx.element_size() * (1 + (dx_add is not None))) + 8)
    with _timed(("hbm", "layernorm_bwd", n0 * n1), nb):
        L.check(lib.dvt_layernorm_bwd_ex(C.byref(q), _stream()), "dvt_layernorm_bwd")
    if defer:
        _ln_deferred.append((pending, defer if callable(defer) else None))
        if len(_ln_deferred) >= LN_DEFER_MAX:
            layernorm_flush()
    return (dx, dg, db) if dx_lp is None else (dx, dg, db, lp)


# ------------------------------------------------------------------ GEMM family
def _gemm_desc(A: Tensor, B: Tensor, M: int, N: int, K: int, *, a_kmajor: bool, b_kmajor: bool, lda: int, ldb: int,
               out: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None, epilogue: int = L.EPI_NONE,
               bias: Optional[Tensor] = None, residual: Optional[Tensor] = None, aux: Optional[Tensor] = None,
               accumulate: bool = False, alpha: float = 1.0, split_k: int = 0, colsum_out: Optional[Tensor] = None,
               colsum_accumulate: bool = False):
    """-> (dvt_gemm_desc, out): argument checks, output allocation and descriptor of one product."""
    _need_cuda(A, B, bias, residual, aux)
    assert A.dtype == B.dtype
    if out_dtype is None:
        out_dtype = A.dtype
    if out is None:
        assert not accumulate
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    assert out.dtype == out_dtype and out.stride(-1) == 1
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    d = L.GemmDesc()
    if residual is not None and residual.dtype != A.dtype:      # fp32 residual stream behind 16-bit operands (dvt_gemm_desc.residual_f32)
        assert residual.dtype == torch.float32 and out_dtype == torch.float32 and epilogue == L.EPI_RESIDUAL
        d.residual_f32 = 1
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, out.stride(0) if out.dim() == 2 else N
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.in_dtype, d.out_dtype = dt(A), _DT[out_dtype]
    d.epilogue, d.accumulate = epilogue, int(accumulate)
    d.bias = _p(bias)
    d.residual = _p(residual)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.aux = _p(aux)
    d.ldaux = aux.stride(0) if aux is not None else 0
    d.alpha = alpha
    d.split_k = split_k
    if colsum_out is not None:
        assert colsum_out.dtype == torch.float32 and colsum_out.numel() == M and colsum_out.is_contiguous()
    d.colsum_out = _p(colsum_out)
    d.colsum_accumulate = int(colsum_accumulate)
    return d, out


def gemm(A: Tensor, B: Tensor, M: int, N: int, K: int, *, a_kmajor: bool, b_kmajor: bool,
         lda: int, ldb: int, out: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None,
         epilogue: int = L.EPI_NONE, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
         aux: Optional[Tensor] = None, accumulate: bool = False, alpha: float = 1.0,
         split_k: int = 0, colsum_out: Optional[Tensor] = None, colsum_accumulate: bool = False,
         defer_reduce: bool = False, carry=None):
    """defer_reduce: a split-K reduce this call would launch is left undone and returned as a pending descriptor --
    (out, pending) -- for the ``carry=`` argument of the NEXT gemm call on this stream (the data gradient of the same
    Linear), which performs it in the idle tail of its own launch; ``out`` is complete only after that call."""
    d, out = _gemm_desc(A, B, M, N, K, a_kmajor=a_kmajor, b_kmajor=b_kmajor, lda=lda, ldb=ldb, out=out, out_dtype=out_dtype,
                        epilogue=epilogue, bias=bias, residual=residual, aux=aux, accumulate=accumulate, alpha=alpha,
                        split_k=split_k, colsum_out=colsum_out, colsum_accumulate=colsum_accumulate)
    out_dtype = out.dtype
    pending = None
    if defer_reduce:
        pending = L.SplitKPending()
        pending._keep = (out, colsum_out)          # the descriptor names their storage
        d.defer_reduce = 1
        d.pending = C.pointer(pending)
    if carry is not None:
        d.carry = C.pointer(carry)
    lib = L.load()
    # the slabs of a deferred reduce must survive the next call: they get a scratch slot of their own
    nws = lib.dvt_gemm_workspace_bytes(C.byref(d))
    ws = _deferred_workspace(nws, A.device, pending) if defer_reduce else workspace(nws, A.device)
    d.workspace = _p(ws)
    if defer_reduce:
        pending._keep += (ws,)
    prof = _profiler
    if prof is not None:
        # algorithmic bytes of the launch: both operands once, every output once, every extra epilogue operand once
        esz, osz = A.element_size(), out.element_size()
        nbytes = esz * (M * K + N * K) + osz * M * N * (2 if accumulate else 1)
        nbytes += esz * M * N * ((residual is not None) + (aux is not None))
        key = ("gemm", int(a_kmajor), int(b_kmajor), M, N, K, epilogue, nbytes)
        prof.begin(key, 2.0 * M * N * K)
        L.check(lib.dvt_gemm(C.byref(d), _stream()), "dvt_gemm")
        prof.end(key)
        return (out, pending) if defer_reduce else out
    L.check(lib.dvt_gemm(C.byref(d), _stream()), "dvt_gemm")
    return (out, pending) if defer_reduce else out


def gemm_is_launch_bound(M: int, N: int, K: int, dtype: torch.dtype, *, a_kmajor: bool = True, b_kmajor: bool = True) -> bool:
    """Does dvt_gemm take the panel-streaming kernel of the launch-bound shapes for this product (no launch)?"""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    d = L.GemmDesc()
    d.A = d.B = d.C = 16                              # (non-null, aligned: the route depends on shapes and layouts only)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = (K if a_kmajor else M), (K if b_kmajor else N), N
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.in_dtype = d.out_dtype = _DT[dtype]
    d.alpha = 1.0
    return L.load().dvt_gemm_route(C.byref(d)) == 0


def splitk_reduce_pending(pending) -> None:
    """Perform a deferred split-K reduce as a launch of its own (no data-gradient launch followed to carry it)."""
    if pending is not None and pending.valid:
        L.check(L.load().dvt_splitk_reduce_pending(C.byref(pending), _stream()), "dvt_splitk_reduce_pending")
        pending.valid = 0


def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, epilogue: int = L.EPI_NONE,
               residual: Optional[Tensor] = None, aux: Optional[Tensor] = None,
               out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """y[M,N] = epi(x[M,K] @ w[N,K]^T + bias).  out_dtype float32 with an fp32 residual: the fp32 residual stream."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.is_contiguous()
    return gemm(x, w, M, N, K, a_kmajor=True, b_kmajor=True, lda=x.stride(0), ldb=K, epilogue=epilogue,
                bias=bias, residual=residual, aux=aux, out_dtype=out_dtype)


def linear_dgrad(dy: Tensor, w: Tensor, *, epilogue: int = L.EPI_NONE, aux: Optional[Tensor] = None,
                 carry=None) -> Tensor:
    """dx[M,K] = epi(dy[M,N] @ w[N,K])   (epilogue DGELU / DRELU multiplies by act'(aux)).
    carry: the pending split-K reduce of the weight gradient launched just before (linear_wgrad(defer_reduce=True))."""
    M, N = dy.shape
    K = w.shape[1]
    assert w.shape[0] == N and dy.stride(1) == 1 and w.is_contiguous()
    out = gemm(dy, w, M, K, N, a_kmajor=True, b_kmajor=False, lda=dy.stride(0), ldb=K, epilogue=epilogue,
               aux=aux, carry=carry)
    if carry is not None:
        carry.valid = 0                              # performed by this call, one way or the other
    return out


def linear_wgrad(dy: Tensor, x: Tensor, *, out: Optional[Tensor] = None, accumulate: bool = False,
                 bias_out: Optional[Tensor] = None, bias_accumulate: bool = False, defer_reduce: bool = False):
    """dW[N,K] (f32) = dy[M,N]^T @ x[M,K]; optionally also the bias gradient
    bias_out[N] (+)= sum_m dy[m, :] from the same pass over dy.  defer_reduce: -> (dW, pending), see ``gemm``."""
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M and dy.stride(1) == 1 and x.stride(1) == 1
    return gemm(dy, x, N, K, M, a_kmajor=False, b_kmajor=False, lda=dy.stride(0), ldb=x.stride(0), out=out,
                out_dtype=torch.float32, accumulate=accumulate, colsum_out=bias_out,
                colsum_accumulate=bias_accumulate, defer_reduce=defer_reduce)


PAIR_LAUNCH = True        # A/B switch (bench.py --no-pair-launch): weight + data gradient of a launch-bound Linear in one launch


def linear_backward(dy: Tensor, x: Tensor, w: Tensor, *, out: Optional[Tensor] = None, accumulate: bool = False,
                    bias_out: Optional[Tensor] = None, bias_accumulate: bool = False, epilogue: int = L.EPI_NONE,
                    aux: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Both gradients of y = x w^T (+ bias) from dy [M,N]: (dW [N,K] f32 (+ bias_out), dx [M,K] = epi(dy w)).
    Launch-bound shapes (a few hundred rows: the temporal encoder, the CLS-row layers): ONE launch (dvt_gemm_pair) --
    the two products are independent.  Otherwise the weight gradient with its split-K reduce deferred, then the data
    gradient carrying that reduce in its grid tail."""
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M and dy.stride(1) == 1 and x.stride(1) == 1 and w.shape == (N, K) and w.is_contiguous()
    wd, out = _gemm_desc(dy, x, N, K, M, a_kmajor=False, b_kmajor=False, lda=dy.stride(0), ldb=x.stride(0), out=out,
                         out_dtype=torch.float32, accumulate=accumulate, colsum_out=bias_out,
                         colsum_accumulate=bias_accumulate)
    gd, _ = _gemm_desc(dy, w, M, K, N, a_kmajor=True, b_kmajor=False, lda=dy.stride(0), ldb=K, out=dy[:, :0].new_empty((0, K)),
                       epilogue=epilogue, aux=aux)
    gd.C, gd.ldc = dy.data_ptr(), K                     # a valid address for the shape test; the real one below
    lib = L.load()
    if PAIR_LAUNCH and lib.dvt_gemm_pair_fused(C.byref(wd), C.byref(gd)):
        dx = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
        gd.C, gd.ldc = dx.data_ptr(), K
        L.check(lib.dvt_gemm_pair(C.byref(wd), C.byref(gd), _stream()), "dvt_gemm_pair")
        return out, dx
    _, pend = linear_wgrad(dy, x, out=out, accumulate=accumulate, bias_out=bias_out, bias_accumulate=bias_accumulate,
                           defer_reduce=True)
    return out, linear_dgrad(dy, w, epilogue=epilogue, aux=aux, carry=pend)


def colsum(x: Tensor, *, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """f32 [N] = sum over rows of x[M,N] (bias gradients)."""
    _need_cuda(x)
    M, N = x.shape
    assert x.stride(1) == 1
    if out is None:
        assert not accumulate
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.numel() == N and out.is_contiguous()
    lib = L.load()
    ws = workspace(lib.dvt_colsum_workspace_bytes(M, N), x.device)
    L.check(lib.dvt_colsum(x.data_ptr(), x.stride(0), out.data_ptr(), _p(ws), M, N, dt(x), int(accumulate),
                           _stream()), "dvt_colsum")
    return out


# ------------------------------------------------------------------ attention
def _attn_desc(q: Tensor, k: Tensor, v: Tensor, o: Tensor, lse: Tensor, scale: float, dropout=None) -> L.AttnDesc:
    """q,k,v,o are 4-D views [B, H, L, dh] with unit stride in the last dim."""
    for t in (q, k, v, o):
        assert t.dim() == 4 and t.stride(3) == 1, "attention operands must be [B,H,L,dh] views, dh contiguous"
    B, H, Lq, dh = q.shape
    Lk = k.shape[2]
    assert k.shape == (B, H, Lk, dh) and v.shape == (B, H, Lk, dh) and o.shape == (B, H, Lq, dh)
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o, d.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    d.B, d.H, d.Lq, d.Lk, d.dh = B, H, Lq, Lk, dh
    d.q_sb, d.q_sh, d.q_sl = q.stride(0), q.stride(1), q.stride(2)
    d.k_sb, d.k_sh, d.k_sl = k.stride(0), k.stride(1), k.stride(2)
    d.v_sb, d.v_sh, d.v_sl = v.stride(0), v.stride(1), v.stride(2)
    d.o_sb, d.o_sh, d.o_sl = o.stride(0), o.stride(1), o.stride(2)
    d.scale = scale
    d.dtype = dt(q)
    if dropout is not None:                      # (p, rng_state tensor, site offset): attention-probability dropout
        d.dropout_p, d.rng_state, d.rng_offset = float(dropout[0]), dropout[1].data_ptr(), int(dropout[2])
    return d


def attention_fwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, scale: float, dropout=None) -> Tensor:
    """Writes o (a [B,H,Lq,dh] view of caller-owned memory); returns lse [B,H,Lq] f32."""
    _need_cuda(q, k, v, o)
    B, H, Lq, _ = q.shape
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, o, lse, scale, dropout)
    nb = (2 * q.numel() + k.numel() + v.numel()) * q.element_size()          # read q, k, v; write o
    with _timed(("hbm", "attention_fwd", q.numel(), 4.0 * q.numel() * k.shape[2]), nb):   # key carries the MFMA flops
        L.check(L.load().dvt_attention_fwd(C.byref(d), _stream()), "dvt_attention_fwd")
    return lse


ATTN_BWD_TWO_PASS = False      # A/B switch (bench.py --attn-two-pass): the dq + dk/dv kernel pair instead of the one-pass backward


def attention_bwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, lse: Tensor, do: Tensor, dq: Tensor,
                  dk: Tensor, dv: Tensor, scale: float, dropout=None, two_pass: Optional[bool] = None) -> None:
    """do must share o's strides; dq/dk/dv must share q/k/v's strides (views of
    caller-owned memory, fully overwritten)."""
    _need_cuda(q, k, v, o, do, dq, dk, dv)
    for a, b in ((do, o), (dq, q), (dk, k), (dv, v)):
        assert a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1), \
            "gradient views must share the layout of their primal"
    d = _attn_desc(q, k, v, o, lse, scale, dropout)
    d.d_o, d.dq, d.dk, d.dv = do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    d.bwd_two_pass = int(ATTN_BWD_TWO_PASS if two_pass is None else two_pass)
    lib = L.load()
    ws = workspace(lib.dvt_attention_bwd_workspace_bytes(C.byref(d)), q.device)
    d.workspace = _p(ws)
    # algorithmic bytes: q, k, v, o, dO read once, dq, dk, dv written once (the one-pass form moves exactly that; the
    # two-kernel form re-reads q, k, v, dO: 13 units through the memory pipeline for these 8)
    nb = (4 * q.numel() + 2 * k.numel() + 2 * v.numel()) * q.element_size()
    with _timed(("hbm", "attention_bwd", q.numel(), 10.0 * q.numel() * k.shape[2]), nb):  # 5 products of 2*L*L*dh
        L.check(lib.dvt_attention_bwd(C.byref(d), _stream()), "dvt_attention_bwd")


# ------------------------------------------------------------------ single-query attention, K / V projections folded
def _attn_cls_desc(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, H: int) -> L.AttnClsDesc:
    """x [S, N, d] (16-bit), last dim contiguous."""
    assert x.dim() == 3 and x.stride(2) == 1
    q = L.AttnClsDesc()
    q.x, q.xs0, q.xs1 = x.data_ptr(), x.stride(0), x.stride(1)
    q.gamma, q.beta, q.eps = gamma.data_ptr(), beta.data_ptr(), eps
    q.S, q.N, q.d = x.shape
    q.H, q.dtype = H, dt(x)
    return q


def attn_cls_supported(x: Tensor, H: int, dh: Optional[int] = None) -> bool:
    """Shape / dtype test of the folded single-query path (no launch): the row-pass kernels (dvt_attn_cls_supported) and,
    when ``dh`` is given, the head-wise products on either side of them (dvt_heads_expand_outer: dh <= 384;
    dvt_heads_contract*: d % 8 == 0, d <= 512) -- so that a shape accepted here cannot fail later inside backward."""
    if not x.is_cuda or x.dim() != 3 or x.stride(2) != 1:
        return False
    if dh is not None and not (0 < dh <= 384 and x.shape[2] % 8 == 0 and x.shape[2] <= 512):
        return False
    q = L.AttnClsDesc()
    q.xs0, q.xs1 = x.stride(0), x.stride(1)
    q.S, q.N, q.d = x.shape
    q.H, q.dtype = H, dt(x)
    return bool(L.load().dvt_attn_cls_supported(C.byref(q)))


def attn_cls_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, R: Tensor):
    """x [S,N,d], R [S,H,d] f32 -> (A [S,H,d] f32, lse [S,H], P [S,N,8], mean [S*N], rstd [S*N])."""
    _need_cuda(x, gamma, beta, R)
    S, N, d = x.shape
    H = R.shape[1]
    assert R.shape == (S, H, d) and R.dtype == torch.float32 and R.is_contiguous()
    q = _attn_cls_desc(x, gamma, beta, eps, H)
    f32 = dict(dtype=torch.float32, device=x.device)
    A, lse, P = torch.empty((S, H, d), **f32), torch.empty((S, H), **f32), torch.empty((S, N, 8), **f32)
    mean, rstd = torch.empty((S * N,), **f32), torch.empty((S * N,), **f32)
    q.R, q.A, q.lse, q.P = R.data_ptr(), A.data_ptr(), lse.data_ptr(), P.data_ptr()
    q.mean, q.rstd = mean.data_ptr(), rstd.data_ptr()
    with _timed(("hbm", "attn_cls_fwd", S * N), S * N * d * x.element_size() + 2 * S * H * d * 4):
        L.check(L.load().dvt_attn_cls_fwd(C.byref(q), _stream()), "dvt_attn_cls_fwd")
    return A, lse, P, mean, rstd


def attn_cls_bwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, R: Tensor, A: Tensor, lse: Tensor, P: Tensor,
                 mean: Tensor, rstd: Tensor, dM: Tensor, *, dx: Optional[Tensor] = None, dg: Optional[Tensor] = None,
                 db: Optional[Tensor] = None, accumulate: bool = False, accumulate_beta: Optional[bool] = None):
    """-> (dx like x, G [S,H,d] f32, dgamma, dbeta)."""
    _need_cuda(x, gamma, beta, R, A, lse, P, mean, rstd, dM)
    S, N, d = x.shape
    H = R.shape[1]
    assert dM.shape == (S, H, d) and dM.dtype == torch.float32 and dM.is_contiguous() and P.shape == (S, N, 8)
    q = _attn_cls_desc(x, gamma, beta, eps, H)
    if dx is None:
        dx = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
    assert dx.stride() == x.stride()
    if dg is None or db is None:
        assert not accumulate and not accumulate_beta and dg is None and db is None
        dg = torch.empty((d,), dtype=torch.float32, device=x.device)
        db = torch.empty((d,), dtype=torch.float32, device=x.device)
    G = torch.empty((S, H, d), dtype=torch.float32, device=x.device)
    q.R, q.A, q.lse, q.P = R.data_ptr(), A.data_ptr(), lse.data_ptr(), P.data_ptr()
    q.mean, q.rstd = mean.data_ptr(), rstd.data_ptr()
    q.dM, q.dx, q.G, q.dgamma, q.dbeta = dM.data_ptr(), dx.data_ptr(), G.data_ptr(), dg.data_ptr(), db.data_ptr()
    q.accumulate_gamma = int(accumulate)
    q.accumulate_beta = int(accumulate if accumulate_beta is None else accumulate_beta)
    lib = L.load()
    ws = workspace(lib.dvt_attn_cls_bwd_workspace_bytes(C.byref(q)), x.device)
    q.workspace = _p(ws)
    with _timed(("hbm", "attn_cls_bwd", S * N), 2 * S * N * d * x.element_size() + 4 * S * H * d * 4):
        L.check(lib.dvt_attn_cls_bwd(C.byref(q), _stream()), "dvt_attn_cls_bwd")
    return dx, G, dg, db


def heads_expand(a: Tensor, W: Tensor, H: int, alpha: float = 1.0) -> Tensor:
    """a [S, H*dh] (16-bit, row stride free), W [H*dh, d] rows of a 16-bit weight -> out[s,h,:] = alpha a[s,h,:] W_h (f32)."""
    _need_cuda(a, W)
    S, inner = a.shape
    d = W.shape[1]
    assert W.shape[0] == inner and a.stride(1) == 1 and W.stride(1) == 1 and a.dtype == W.dtype and inner % H == 0
    out = torch.empty((S, H, d), dtype=torch.float32, device=a.device)
    L.check(L.load().dvt_heads_expand(a.data_ptr(), a.stride(0), W.data_ptr(), W.stride(0), out.data_ptr(), S, H,
                                      inner // H, d, alpha, dt(a), _stream()), "dvt_heads_expand")
    return out


def heads_contract(v: Tensor, W: Tensor, alpha: float = 1.0, gamma: Optional[Tensor] = None,
                   beta: Optional[Tensor] = None) -> Tensor:
    """v [S,H,d] f32 (entering as gamma*v+beta), W [H*dh, d] 16-bit -> out [S, H*dh] = alpha W_h v_h in W's dtype."""
    _need_cuda(v, W, gamma, beta)
    S, H, d = v.shape
    inner = W.shape[0]
    assert W.shape[1] == d and W.stride(1) == 1 and v.is_contiguous() and v.dtype == torch.float32 and inner % H == 0
    out = torch.empty((S, inner), dtype=W.dtype, device=v.device)
    L.check(L.load().dvt_heads_contract(v.data_ptr(), _p(gamma), _p(beta), W.data_ptr(), W.stride(0), out.data_ptr(),
                                        inner, S, H, inner // H, d, alpha, dt(W), _stream()), "dvt_heads_contract")
    return out


def heads_outer(a: Tensor, v: Tensor, out: Tensor, alpha: float = 1.0, gamma: Optional[Tensor] = None,
                beta: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """out[h*dh+e, c] (+)= alpha sum_s a[s, h*dh+e] (gamma*v+beta)[s,h,c]; a [S, H*dh] 16-bit, v [S,H,d] f32,
    out [H*dh, d] f32 (row stride free): the weight gradient of a row range of the packed to_qkv."""
    _need_cuda(a, v, out, gamma, beta)
    S, H, d = v.shape
    inner = a.shape[1]
    assert out.shape == (inner, d) and out.stride(1) == 1 and out.dtype == torch.float32 and a.stride(1) == 1
    assert v.is_contiguous() and v.dtype == torch.float32 and inner % H == 0
    L.check(L.load().dvt_heads_outer(a.data_ptr(), a.stride(0), v.data_ptr(), _p(gamma), _p(beta), out.data_ptr(),
                                     out.stride(0), S, H, inner // H, d, alpha, int(accumulate), dt(a), _stream()),
            "dvt_heads_outer")
    return out


def heads_expand_outer(a: Tensor, W: Tensor, H: int, v: Tensor, dW: Tensor, *, alpha_out: float = 1.0, alpha_dw: float = 1.0,
                       gamma: Optional[Tensor] = None, beta: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """One launch: ``heads_expand(a, W, H, alpha_out)`` (returned) and ``heads_outer(a, v, dW, alpha_dw, gamma, beta)``."""
    _need_cuda(a, W, v, dW, gamma, beta)
    S, inner = a.shape
    d = W.shape[1]
    assert W.shape[0] == inner and a.stride(1) == 1 and W.stride(1) == 1 and a.dtype == W.dtype and inner % H == 0
    assert v.shape == (S, H, d) and v.is_contiguous() and v.dtype == torch.float32
    assert dW.shape == (inner, d) and dW.stride(1) == 1 and dW.dtype == torch.float32
    out = torch.empty((S, H, d), dtype=torch.float32, device=a.device)
    L.check(L.load().dvt_heads_expand_outer(a.data_ptr(), a.stride(0), W.data_ptr(), W.stride(0), out.data_ptr(), alpha_out,
                                            v.data_ptr(), _p(gamma), _p(beta), dW.data_ptr(), dW.stride(0), alpha_dw,
                                            int(accumulate), S, H, inner // H, d, dt(a), _stream()), "dvt_heads_expand_outer")
    return out


def heads_contract_outer(v: Tensor, gamma: Tensor, W: Tensor, a: Tensor, dW: Tensor, *, alpha_out: float = 1.0,
                         alpha_dw: float = 1.0, accumulate: bool = False) -> Tensor:
    """One launch: ``heads_contract(v, W, alpha_out, gamma)`` (returned) and ``heads_outer(a, v, dW, alpha_dw, gamma)``."""
    _need_cuda(v, gamma, W, a, dW)
    S, H, d = v.shape
    inner = W.shape[0]
    assert W.shape[1] == d and W.stride(1) == 1 and v.is_contiguous() and v.dtype == torch.float32 and inner % H == 0
    assert a.shape == (S, inner) and a.stride(1) == 1 and a.dtype == W.dtype
    assert dW.shape == (inner, d) and dW.stride(1) == 1 and dW.dtype == torch.float32
    out = torch.empty((S, inner), dtype=W.dtype, device=v.device)
    L.check(L.load().dvt_heads_contract_outer(v.data_ptr(), gamma.data_ptr(), W.data_ptr(), W.stride(0), out.data_ptr(), inner,
                                              alpha_out, a.data_ptr(), a.stride(0), dW.data_ptr(), dW.stride(0), alpha_dw,
                                              int(accumulate), S, H, inner // H, d, dt(W), _stream()),
            "dvt_heads_contract_outer")
    return out


# ------------------------------------------------------------------ losses / optimizer
def bce_logits_fwd(z: Tensor, target: Tensor) -> Tensor:
    _need_cuda(z, target)
    assert z.is_contiguous() and target.is_contiguous() and target.dtype == torch.float32
    loss = torch.empty((1,), dtype=torch.float32, device=z.device)
    L.check(L.load().dvt_bce_logits_fwd(z.data_ptr(), target.data_ptr(), loss.data_ptr(), z.numel(), dt(z),
                                        _stream()), "dvt_bce_logits_fwd")
    return loss


def head_bce_supported(rows: int, d: int, classes: int) -> bool:
    return bool(L.load().dvt_head_bce_supported(int(rows), int(d), int(classes)))


def head_bce_fwd(x: Tensor, g1, b1, eps1: float, g2: Tensor, b2: Tensor, eps2: float, w: Tensor, c, target: Tensor):
    """LN -> LN -> Linear -> mean BCE in one launch (dvt_head_bce_fwd) -> (loss [1], logits [rows, C], grads): grads maps
    "x", "g1", "b1", "g2", "b2", "w", "c" to fp32 views holding d(loss)/d(.) for an upstream gradient of 1."""
    _need_cuda(x, g2, b2, w, target)
    rows, d = x.shape
    Cc = w.shape[0]
    assert x.is_contiguous() and w.is_contiguous() and w.shape[1] == d and target.shape == (rows, Cc)
    for t in (g1, b1, g2, b2, w, c, target):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    lib = L.load()
    n = lib.dvt_head_bce_grads_elems(rows, d, Cc)
    if n < 0:
        raise ValueError(f"head_bce_fwd: unsupported shape rows={rows} d={d} classes={Cc}")
    G = torch.empty((n,), dtype=torch.float32, device=x.device)
    logits = torch.empty((rows, Cc), dtype=torch.float32, device=x.device)
    loss = torch.empty((1,), dtype=torch.float32, device=x.device)
    desc = L.HeadBceDesc()
    desc.x, desc.x_dtype = x.data_ptr(), dt(x)
    desc.g1, desc.b1 = (g1.data_ptr(), b1.data_ptr()) if g1 is not None else (None, None)
    desc.g2, desc.b2, desc.w, desc.c = g2.data_ptr(), b2.data_ptr(), w.data_ptr(), (c.data_ptr() if c is not None else None)
    desc.target, desc.logits, desc.loss, desc.grads = target.data_ptr(), logits.data_ptr(), loss.data_ptr(), G.data_ptr()
    desc.rows, desc.d, desc.classes, desc.eps1, desc.eps2 = rows, d, Cc, float(eps1), float(eps2)
    L.check(lib.dvt_head_bce_fwd(C.byref(desc), _stream()), "dvt_head_bce_fwd")
    o, grads = 0, {}
    for name, cnt in (("x", rows * d), ("g1", d), ("b1", d), ("g2", d), ("b2", d), ("w", Cc * d), ("c", (Cc + 63) // 64 * 64)):
        grads[name] = G[o:o + (Cc if name == "c" else cnt)]
        o += cnt
    grads["x"] = grads["x"].view(rows, d)
    grads["w"] = grads["w"].view(Cc, d)
    return loss, logits, grads


def scaled_emit_group(scale: Tensor, entries) -> None:
    """dst = scale * src (+ dst), optional 16-bit copy of the result; entries: (src f32, dst f32 or None, accumulate,
    dst_lp 16-bit or None).  One launch (dvt_scaled_emit_group)."""
    if not entries:
        return
    _need_cuda(scale)
    assert scale.dtype == torch.float32 and scale.numel() == 1
    arr = (L.EmitEntry * len(entries))()
    for i, (src, dst, acc, lp) in enumerate(entries):
        _need_cuda(src)
        assert src.dtype == torch.float32 and src.is_contiguous()
        assert dst is None or (dst.dtype == torch.float32 and dst.is_contiguous() and dst.numel() == src.numel())
        assert lp is None or (lp.is_contiguous() and lp.numel() == src.numel())
        e = arr[i]
        e.src, e.dst, e.dst_lp = src.data_ptr(), (dst.data_ptr() if dst is not None else None), (lp.data_ptr() if lp is not None else None)
        e.n, e.accumulate, e.lp_dtype = src.numel(), int(bool(acc)), (dt(lp) if lp is not None else 0)
    L.check(L.load().dvt_scaled_emit_group(scale.data_ptr(), C.cast(arr, C.c_void_p), len(entries), _stream()),
            "dvt_scaled_emit_group")


def bce_logits_bwd(z: Tensor, target: Tensor, gloss: Tensor) -> Tensor:
    _need_cuda(z, target, gloss)
    dz = torch.empty_like(z)
    L.check(L.load().dvt_bce_logits_bwd(z.data_ptr(), target.data_ptr(), gloss.data_ptr(), dz.data_ptr(), z.numel(),
                                        dt(z), _stream()), "dvt_bce_logits_bwd")
    return dz


def ce_argmax_fwd(student: Tensor, teacher: Tensor) -> Tensor:
    _need_cuda(student, teacher)
    assert student.is_contiguous() and teacher.is_contiguous() and student.dtype == teacher.dtype
    rows, Cn = student.shape
    loss = torch.empty((1,), dtype=torch.float32, device=student.device)
    L.check(L.load().dvt_ce_argmax_fwd(student.data_ptr(), teacher.data_ptr(), loss.data_ptr(), rows, Cn,
                                       dt(student), _stream()), "dvt_ce_argmax_fwd")
    return loss


def ce_argmax_bwd(student: Tensor, teacher: Tensor, gloss: Tensor) -> Tensor:
    rows, Cn = student.shape
    ds = torch.empty_like(student)
    L.check(L.load().dvt_ce_argmax_bwd(student.data_ptr(), teacher.data_ptr(), gloss.data_ptr(), ds.data_ptr(), rows,
                                       Cn, dt(student), _stream()), "dvt_ce_argmax_bwd")
    return ds


def adamw_step_(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, *, lr: float, beta1: float,
                beta2: float, eps: float, weight_decay: float, step: int) -> None:
    _need_cuda(param, grad, exp_avg, exp_avg_sq)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous()
    L.check(L.load().dvt_adamw_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                    param.numel(), lr, beta1, beta2, eps, weight_decay, step, _stream()),
            "dvt_adamw_step")


def frames_preprocess(frames: Tensor, resize: int, crop: int, mean, std, out_dtype: torch.dtype = torch.bfloat16) -> Tensor:
    """uint8 RGB frames [F, H0, W0, 3] -> [F, 3, crop, crop]: Resize(resize) + CenterCrop(crop) + ToTensor +
    Normalize(mean, std), bit-exact with PIL / torch in fp32 (MMX_Light_dl.py:203-217)."""
    _need_cuda(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise ValueError("frames must be uint8 [F, H0, W0, 3]")
    frames = frames.contiguous()
    Fr, H0, W0, _ = frames.shape
    lib = L.load()
    nbytes = lib.dvt_frames_preprocess_workspace_bytes(Fr, H0, W0, resize, crop)
    if nbytes == 0:
        raise ValueError(f"crop {crop} exceeds the frame resized to shorter side {resize} ({H0}x{W0})")
    ws = workspace(nbytes, frames.device)
    out = torch.empty((Fr, 3, crop, crop), dtype=out_dtype, device=frames.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    L.check(lib.dvt_frames_preprocess(frames.data_ptr(), out.data_ptr(), _DT[out_dtype], Fr, H0, W0, resize, crop,
                                      C.cast(m, C.c_void_p), C.cast(sd, C.c_void_p), ws.data_ptr(),
                                      _stream()), "dvt_frames_preprocess")
    return out


def f1_samples(probs: Tensor, labels: Tensor, thresholds) -> Tensor:
    """samples-averaged F1 of ``probs > t`` for every t (callbacks.py:38-41) -> f32 [T] on the device."""
    _need_cuda(probs, labels)
    probs = probs.contiguous() if probs.dtype == torch.float32 else cast(probs.contiguous(), torch.float32)
    lab = labels.contiguous()
    if lab.dtype != torch.uint8:
        lab = (lab != 0).to(torch.uint8)          # dtype plumbing of a label mask
    N, Cn = probs.shape
    th = [float(t) for t in thresholds]
    lib = L.load()
    ws = workspace(lib.dvt_f1_samples_workspace_bytes(N, len(th)), probs.device)
    out = torch.empty((len(th),), dtype=torch.float32, device=probs.device)
    arr = (C.c_float * len(th))(*th)
    L.check(lib.dvt_f1_samples(probs.data_ptr(), lab.data_ptr(), N, Cn, C.cast(arr, C.c_void_p), len(th),
                               out.data_ptr(), ws.data_ptr(), _stream()), "dvt_f1_samples")
    return out


def average_precision(probs: Tensor, labels: Tensor):
    """-> (samples-averaged AP [1], support-weighted AP [1], per-class AP [C]) f32 device tensors
    (``average_precision_score`` of callbacks.py:48-54)."""
    _need_cuda(probs, labels)
    probs = probs.contiguous() if probs.dtype == torch.float32 else cast(probs.contiguous(), torch.float32)
    lab = labels.contiguous()
    if lab.dtype != torch.uint8:
        lab = (lab != 0).to(torch.uint8)
    N, Cn = probs.shape
    lib = L.load()
    nbytes = lib.dvt_average_precision_workspace_bytes(N, Cn)
    if nbytes == 0:
        raise ValueError("average_precision: empty input or N*C >= 2^31")
    ws = workspace(nbytes, probs.device)
    out = torch.empty((2 + Cn,), dtype=torch.float32, device=probs.device)
    L.check(lib.dvt_average_precision(probs.data_ptr(), lab.data_ptr(), N, Cn, out.data_ptr(), out[1:].data_ptr(),
                                      out[2:].data_ptr(), ws.data_ptr(), _stream()), "dvt_average_precision")
    return out[0:1], out[1:2], out[2:]


def l2norm_rows_fwd(x: Tensor, eps: float):
    _need_cuda(x)
    x = x.contiguous()
    rows, D = x.shape
    y = torch.empty_like(x)
    inv = torch.empty((rows,), dtype=torch.float32, device=x.device)
    L.check(L.load().dvt_l2norm_rows_fwd(x.data_ptr(), y.data_ptr(), inv.data_ptr(), rows, D, eps, dt(x), _stream()),
            "dvt_l2norm_rows_fwd")
    return y, inv


def l2norm_rows_bwd(dy: Tensor, y: Tensor, inv: Tensor, eps: float) -> Tensor:
    dy = dy.contiguous()
    rows, D = y.shape
    dx = torch.empty_like(y)
    L.check(L.load().dvt_l2norm_rows_bwd(dy.data_ptr(), y.data_ptr(), inv.data_ptr(), dx.data_ptr(), rows, D, eps, dt(y),
                                         _stream()), "dvt_l2norm_rows_bwd")
    return dx


def cosine_rows(a: Tensor, b: Tensor, eps: float = 1e-8) -> Tensor:
    """nn.CosineSimilarity(dim=1): [rows, D] x [rows, D] -> f32 [rows] (no gradient)."""
    _need_cuda(a, b)
    a, b = a.detach().contiguous(), b.detach().contiguous()
    assert a.shape == b.shape and a.dim() == 2 and a.dtype == b.dtype
    out = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device)
    L.check(L.load().dvt_cosine_rows(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1], eps, dt(a),
                                     _stream()), "dvt_cosine_rows")
    return out


def gate_fwd(a: Tensor, b: Tensor) -> Tensor:
    _need_cuda(a, b)
    a, b = a.contiguous(), b.contiguous()
    assert a.shape == b.shape and a.dtype == b.dtype
    y = torch.empty_like(a)
    L.check(L.load().dvt_gate_fwd(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), dt(a), _stream()), "dvt_gate_fwd")
    return y


def gate_bwd(dy: Tensor, a: Tensor, b: Tensor):
    dy = dy.contiguous()
    da, db = torch.empty_like(a), torch.empty_like(b)
    L.check(L.load().dvt_gate_bwd(dy.data_ptr(), a.data_ptr(), b.data_ptr(), da.data_ptr(), db.data_ptr(), a.numel(), dt(a),
                                  _stream()), "dvt_gate_bwd")
    return da, db


def contrastive_fwd(sim: Tensor, temperature: float):
    _need_cuda(sim)
    M = sim.shape[0]
    assert sim.dtype == torch.float32 and sim.is_contiguous() and sim.shape == (M, M)
    buf = torch.empty((2 * M + 1,), dtype=torch.float32, device=sim.device)
    L.check(L.load().dvt_contrastive_fwd(sim.data_ptr(), M, temperature, buf[2 * M:].data_ptr(), buf.data_ptr(),
                                         buf[M:].data_ptr(), _stream()), "dvt_contrastive_fwd")
    return buf[2 * M:].view(()), buf[:M]                     # loss, row_lse


def contrastive_bwd(sim: Tensor, row_lse: Tensor, temperature: float, gloss: Tensor) -> Tensor:
    M = sim.shape[0]
    dsim = torch.empty_like(sim)
    L.check(L.load().dvt_contrastive_bwd(sim.data_ptr(), row_lse.data_ptr(), M, temperature, gloss.data_ptr(),
                                         dsim.data_ptr(), _stream()), "dvt_contrastive_bwd")
    return dsim


def adamw_step_scaled_(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step_dev: Tensor, scale: Tensor,
                       found_inf: Tensor, good_steps: Tensor, loss_grad: Tensor, *, lr: float, beta1: float, beta2: float,
                       eps: float, weight_decay: float, growth_interval: int, growth: float, backoff: float,
                       loss_grad_base: float, skip: Optional[Tensor] = None) -> None:
    """AdamW under dynamic loss scaling, all scaler state on the device (hipGraph-capturable)."""
    _need_cuda(param, grad, exp_avg, exp_avg_sq, step_dev, scale, found_inf, good_steps, loss_grad)
    assert found_inf.dtype == torch.int32 and good_steps.dtype == torch.int32 and step_dev.dtype == torch.int64
    L.check(L.load().dvt_adamw_step_scaled(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                           param.numel(), lr, beta1, beta2, eps, weight_decay, step_dev.data_ptr(),
                                           scale.data_ptr(), found_inf.data_ptr(), good_steps.data_ptr(), growth_interval,
                                           growth, backoff, loss_grad.data_ptr(), loss_grad_base, _skip(skip, param),
                                           _stream()),
            "dvt_adamw_step_scaled")


def device_delay(microseconds: int) -> None:
    """Keep the current stream busy for a while (measurement aid, see dvt_device_delay)."""
    L.check(L.load().dvt_device_delay(int(microseconds), _stream()), "dvt_device_delay")


def dropout(x: Tensor, p: float, rng_state: Tensor, call_offset: int) -> Tensor:
    """y = x * keep / (1 - p); keep is a pure function of (rng_state on the device, call_offset, element index)."""
    _need_cuda(x, rng_state)
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.load().dvt_dropout(x.data_ptr(), y.data_ptr(), x.numel(), p, rng_state.data_ptr(), call_offset, dt(x),
                                 _stream()), "dvt_dropout")
    return y


def dropout_fused(x: Tensor, p: float, rng_state: Optional[Tensor], call_offset: int, *, residual: Optional[Tensor] = None,
                  relu: bool = False, gate: Optional[Tensor] = None) -> Tensor:
    """y = residual? + keep * relu?(x) / (1 - p) with the mask of ``dropout`` at (rng_state, call_offset); or, with ``gate``
    (the forward's output) and no rng_state, the backward of the relu form: gate != 0 ? x / (1 - p) : 0 (dvt_dropout_fused)."""
    _need_cuda(x, residual, gate, rng_state)
    x = x.contiguous()
    assert (rng_state is None) != (gate is None)
    for t in (residual, gate):
        assert t is None or (t.shape == x.shape and t.dtype == x.dtype and t.is_contiguous())
    y = torch.empty_like(x)
    L.check(L.load().dvt_dropout_fused(x.data_ptr(), _p(residual), _p(gate), y.data_ptr(), x.numel(), p, _p(rng_state),
                                       call_offset, int(relu), dt(x), _stream()), "dvt_dropout_fused")
    return y


def rng_advance_(rng_state: Tensor, delta: int) -> None:
    L.check(L.load().dvt_rng_advance(rng_state.data_ptr(), delta, _stream()), "dvt_rng_advance")


def _skip(skip: Optional[Tensor], param: Tensor):
    """Pointer of the optional 64-element-block skip mask of the flat-buffer optimizer steps."""
    if skip is None:
        return None
    _need_cuda(skip)
    assert skip.dtype == torch.uint8 and skip.is_contiguous() and skip.numel() * 64 >= param.numel()
    return skip.data_ptr()


def sgd_step_(param: Tensor, grad: Tensor, momentum_buf: Optional[Tensor], *, lr: float, momentum: float,
              weight_decay: float, skip: Optional[Tensor] = None) -> None:
    _need_cuda(param)
    assert param.dtype == grad.dtype == torch.float32 and param.is_contiguous() and grad.is_contiguous()
    L.check(L.load().dvt_sgd_step(param.data_ptr(), grad.data_ptr(), _p(momentum_buf), param.numel(), lr, momentum,
                                  weight_decay, _skip(skip, param), _stream()), "dvt_sgd_step")


def adagrad_step_(param: Tensor, grad: Tensor, state_sum: Tensor, *, lr: float, lr_decay: float, eps: float,
                  weight_decay: float, step: int, skip: Optional[Tensor] = None) -> None:
    _need_cuda(param)
    assert param.dtype == grad.dtype == torch.float32 and param.is_contiguous() and grad.is_contiguous()
    L.check(L.load().dvt_adagrad_step(param.data_ptr(), grad.data_ptr(), state_sum.data_ptr(), param.numel(), lr,
                                      lr_decay, eps, weight_decay, step, _skip(skip, param), _stream()), "dvt_adagrad_step")


def adamw_step_dev_(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step_dev: Tensor, *,
                    lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
                    skip: Optional[Tensor] = None) -> None:
    """AdamW with the step counter on the device (hipGraph-capturable)."""
    _need_cuda(param, grad, exp_avg, exp_avg_sq, step_dev)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert step_dev.dtype == torch.int64 and step_dev.numel() == 1
    L.check(L.load().dvt_adamw_step_dev(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(),
                                        exp_avg_sq.data_ptr(), param.numel(), lr, beta1, beta2, eps,
                                        weight_decay, step_dev.data_ptr(), _skip(skip, param), _stream()),
            "dvt_adamw_step_dev")


def adamw_step_fused_(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step_dev2: Tensor, *,
                      lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
                      skip: Optional[Tensor] = None, mirror: Optional[Tensor] = None) -> None:
    """AdamW + the 16-bit weight mirror + the device step counter (int64[2]: steps, ticket) in one launch."""
    _need_cuda(param, grad, exp_avg, exp_avg_sq, step_dev2, mirror)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert step_dev2.dtype == torch.int64 and step_dev2.numel() == 2
    if mirror is not None:
        assert mirror.numel() == param.numel() and mirror.is_contiguous() and mirror.dtype in (torch.bfloat16, torch.float16)
    L.check(L.load().dvt_adamw_step_fused(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                          param.numel(), lr, beta1, beta2, eps, weight_decay, step_dev2.data_ptr(),
                                          _skip(skip, param), _p(mirror), _DT[mirror.dtype] if mirror is not None else 0,
                                          _stream()), "dvt_adamw_step_fused")


# ------------------------------------------------------------------ per-frame CNN encoder (csrc/conv.hip)
def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def conv_out_hw(H: int, W: int, k, stride, pad) -> Tuple[int, int]:
    (kh, kw), (sh, sw), (ph, pw) = _pair(k), _pair(stride), _pair(pad)
    return (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1


def im2col(x: Tensor, nchw: bool, N: int, Cc: int, H: int, W: int, k, stride, pad, ld: int,
           out_dtype: torch.dtype) -> Tensor:
    """k / stride / pad: int or (h, w) pair."""
    _need_cuda(x)
    assert x.is_contiguous()
    (kh, kw), (sh, sw), (ph, pw) = _pair(k), _pair(stride), _pair(pad)
    Ho, Wo = conv_out_hw(H, W, k, stride, pad)
    out = torch.empty((N * Ho * Wo, ld), dtype=out_dtype, device=x.device)
    L.check(L.load().dvt_im2col(x.data_ptr(), dt(x), int(nchw), out.data_ptr(), _DT[out_dtype], N, Cc, H, W, kh, kw,
                                sh, sw, ph, pw, ld, _stream()), "dvt_im2col")
    return out


def col2im(dcol: Tensor, N: int, Cc: int, H: int, W: int, k, stride, pad, *, add: Optional[Tensor] = None,
           add_stride: int = 0) -> Tensor:
    """add: a second gradient path into the same map, summed in the same pass -- [N*H*W, C] (add_stride 0), or the compact
    gradient of the map's stride-``add_stride`` subsampling, added at the pixels that subsampling reads."""
    _need_cuda(dcol, add)
    assert dcol.is_contiguous()
    (kh, kw), (sh, sw), (ph, pw) = _pair(k), _pair(stride), _pair(pad)
    dx = torch.empty((N * H * W, Cc), dtype=dcol.dtype, device=dcol.device)
    if add is not None:
        s_ = add_stride
        rows = N * H * W if s_ == 0 else N * ((H + s_ - 1) // s_) * ((W + s_ - 1) // s_)
        assert add.is_contiguous() and add.dtype == dcol.dtype and tuple(add.shape) == (rows, Cc)
    L.check(L.load().dvt_col2im(dcol.data_ptr(), dx.data_ptr(), N, Cc, H, W, kh, kw, sh, sw, ph, pw, dcol.shape[1],
                                _p(add), add_stride, dt(dcol), _stream()), "dvt_col2im")
    return dx


def col2im_nchw(dcol: Tensor, N: int, Cc: int, H: int, W: int, k, stride, pad, dx_dtype: torch.dtype) -> Tensor:
    """Adjoint gather written as an NCHW tensor [N, C, H, W] in ``dx_dtype`` (stem input gradient)."""
    _need_cuda(dcol)
    assert dcol.is_contiguous()
    (kh, kw), (sh, sw), (ph, pw) = _pair(k), _pair(stride), _pair(pad)
    dx = torch.empty((N, Cc, H, W), dtype=dx_dtype, device=dcol.device)
    L.check(L.load().dvt_col2im_nchw(dcol.data_ptr(), dt(dcol), dx.data_ptr(), _DT[dx_dtype], N, Cc, H, W, kh, kw, sh,
                                     sw, ph, pw, dcol.shape[1], _stream()), "dvt_col2im_nchw")
    return dx


def _conv_desc(x: Tensor, wp: Tensor, y, N, Cc, H, W, Cout, k, stride, pad, trim_w: int = 0):
    (kh, kw), (sh, sw), (ph, pw) = _pair(k), _pair(stride), _pair(pad)
    d = L.ConvDesc()
    d.x, d.w, d.y = x.data_ptr(), wp.data_ptr(), (y.data_ptr() if y is not None else 1 << 4)
    d.N, d.H, d.W, d.C, d.Cout = N, H, W, Cc, Cout
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw = kh, kw, sh, sw, ph, pw
    d.dtype = dt(x)
    d.trim_w = trim_w
    return d


def conv2d_implicit_k(Cc: int, Cout: int, k) -> int:
    """Row length of the packed weights the implicit forward expects (kh*kw*C; rounded up to the k-tile for C == 8)."""
    (kh, kw) = _pair(k)
    d = L.ConvDesc()
    d.C, d.Cout, d.kh, d.kw = Cc, Cout, kh, kw
    return int(L.load().dvt_conv2d_implicit_k(C.byref(d)))


def conv2d_implicit_supported(x: Tensor, wp: Tensor, N, Cc, H, W, Cout, k, stride, pad, trim_w: int = 0) -> bool:
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or wp.dtype != x.dtype:
        return False
    if not (x.is_contiguous() and wp.is_contiguous() and wp.shape == (Cout, conv2d_implicit_k(Cc, Cout, k))):
        return False
    return bool(L.load().dvt_conv2d_implicit_supported(C.byref(_conv_desc(x, wp, None, N, Cc, H, W, Cout, k, stride, pad, trim_w))))


def nchw_to_nhwc_pad(x: Tensor, dtype: torch.dtype, cpad: int = 8) -> Tensor:
    """Raw frames [N, C, H, W] -> the stem's input form in ``dtype``: cpad 8 = NHWC matrix [N*H*W, 8] (channels C.. zero);
    cpad 4 (C <= 4, W even) = [N*H*(W/2), 8], one row per pair of horizontally adjacent pixels (4 channels each)."""
    _need_cuda(x)
    x = x.contiguous()
    N, Cc, H, W = x.shape
    y = torch.empty((N * H * W * cpad // 8, 8), dtype=dtype, device=x.device)
    L.check(L.load().dvt_nchw_to_nhwc_pad(x.data_ptr(), dt(x), y.data_ptr(), _DT[dtype], N, Cc, H, W, cpad, _stream()),
            "dvt_nchw_to_nhwc_pad")
    return y


def conv_weight_pairs(w: Tensor, Cout: int, Cin: int, kh: int, kw: int, pw: int, kwp: int) -> Tensor:
    """Stem weights f32 [Cout, Cin <= 4, kh, kw] -> pixel-pair form f32 [Cout, 8, kh, kwp] (dvt_conv_weight_pairs)."""
    _need_cuda(w)
    w = w.detach().contiguous()
    assert w.dtype == torch.float32 and w.numel() == Cout * Cin * kh * kw
    out = torch.empty((Cout, 8, kh, kwp), dtype=torch.float32, device=w.device)
    L.check(L.load().dvt_conv_weight_pairs(w.data_ptr(), out.data_ptr(), Cout, Cin, kh, kw, pw, kwp, _stream()),
            "dvt_conv_weight_pairs")
    return out


def conv_weight_pairs_bwd(dwp: Tensor, Cout: int, Cin: int, kh: int, kw: int, pw: int, kwp: int, *,
                          out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """Adjoint: f32 [Cout, 8, kh, kwp] -> f32 [Cout, Cin, kh, kw] (+= into ``out`` when accumulate)."""
    _need_cuda(dwp)
    assert dwp.dtype == torch.float32 and dwp.is_contiguous() and dwp.numel() == Cout * 8 * kh * kwp
    if out is None:
        assert not accumulate
        out = torch.empty((Cout, Cin, kh, kw), dtype=torch.float32, device=dwp.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == Cout * Cin * kh * kw
    L.check(L.load().dvt_conv_weight_pairs_bwd(dwp.data_ptr(), out.data_ptr(), Cout, Cin, kh, kw, pw, kwp, int(accumulate),
                                               _stream()), "dvt_conv_weight_pairs_bwd")
    return out


def conv_stem7_supported(xp: Tensor, wp: Tensor, N: int, H: int, Wp: int) -> bool:
    """Does dvt_conv_stem7 take this stem?  xp: the pixel-pair map [N*H*Wp, 8] (nchw_to_nhwc_pad(.., 4)), wp [64, >= 224]."""
    if not xp.is_cuda or xp.dtype not in (torch.bfloat16, torch.float16) or wp.dtype != xp.dtype:
        return False
    if not (xp.is_contiguous() and wp.is_contiguous() and xp.numel() == N * H * Wp * 8 and wp.dim() == 2 and wp.shape[0] == 64
            and wp.shape[1] >= 224 and wp.shape[1] % 8 == 0):
        return False
    return bool(L.load().dvt_conv_stem7_supported(N, H, Wp, dt(xp)))


def conv_stem7(xp: Tensor, wp: Tensor, N: int, H: int, Wp: int, want_stats: bool = False):
    """The 7x7 / 2 / 3 stem on the pixel-pair map from an LDS halo patch, weights in registers (dvt_conv_stem7): same result
    and statistics contract as conv2d_implicit with the pair geometry ((7, 4) / (2, 1) / (3, 2), trim_w = 1).
    -> z [N * (H/2) * Wp, 64] (, partial, parts)."""
    _need_cuda(xp, wp)
    Ho = H // 2
    y = torch.empty((N * Ho * Wp, 64), dtype=xp.dtype, device=xp.device)
    lib = L.load()
    partial, parts = None, 0
    if want_stats:
        parts = int(lib.dvt_conv_stem7_stats_parts(N, H, Wp))
        partial = workspace((parts + 64) * 2 * 64 * 4, xp.device, slot="bn_partial")
    nb = (xp.numel() + y.numel() + wp.numel()) * xp.element_size()
    with _timed(("conv", "stem7", N * Ho * Wp, 64, 224, nb), 2.0 * N * Ho * Wp * 64 * 224):
        L.check(lib.dvt_conv_stem7(xp.data_ptr(), wp.data_ptr(), wp.shape[1], y.data_ptr(), _p(partial), N, H, Wp, dt(xp),
                                   _stream()), "dvt_conv_stem7")
    return (y, partial, parts) if want_stats else y


def conv3x3_c64_supported(x: Tensor, wp: Tensor, N: int, H: int, W: int) -> bool:
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or wp.dtype != x.dtype:
        return False
    if not (x.is_contiguous() and wp.is_contiguous() and tuple(wp.shape) == (64, 576) and x.shape == (N * H * W, 64)):
        return False
    return bool(L.load().dvt_conv3x3_c64_supported(N, H, W, dt(x)))


def conv3x3_c64(x: Tensor, wp: Tensor, N: int, H: int, W: int, want_stats: bool = False, residual: Optional[Tensor] = None):
    """3x3 / 1 / 1 convolution, 64 -> 64 channels, from an LDS-resident halo patch (dvt_conv3x3_c64); same contract as
    conv2d_implicit (residual: added to the output)."""
    _need_cuda(x, wp, residual)
    y = torch.empty((N * H * W, 64), dtype=x.dtype, device=x.device)
    if residual is not None:
        assert residual.is_contiguous() and residual.dtype == x.dtype and residual.shape == y.shape
    lib = L.load()
    partial, parts = None, 0
    if want_stats:
        parts = int(lib.dvt_conv3x3_c64_stats_parts(N, H, W))
        partial = workspace((parts + 64) * 2 * 64 * 4, x.device, slot="bn_partial")
    with _timed(("conv", "halo3x3_c64", N * H * W, 64, 576, (2 * x.numel() + wp.numel()) * x.element_size()), 2.0 * N * H * W * 64 * 576):
        L.check(lib.dvt_conv3x3_c64(x.data_ptr(), wp.data_ptr(), y.data_ptr(), _p(partial), _p(residual), N, H, W, dt(x),
                                    _stream()), "dvt_conv3x3_c64")
    return (y, partial, parts) if want_stats else y


# (DVT_STREAM_LAYER2=0: layer 2's 128 <-> 288 pairs stay on the implicit GEMM -- the same-box A/B switch of round 6)
_STREAM_PAIRS = ((64, 144), (144, 64)) + (((128, 288), (288, 128)) if os.environ.get("DVT_STREAM_LAYER2", "1") != "0" else ())


def conv3x3_stream_supported(x: Tensor, wp: Tensor, N: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or wp.dtype != x.dtype:
        return False
    if (Cin, Cout) not in _STREAM_PAIRS:
        return False
    if not (x.is_contiguous() and wp.is_contiguous() and tuple(wp.shape) == (Cout, 9 * Cin) and x.shape == (N * H * W, Cin)):
        return False
    return bool(L.load().dvt_conv3x3_stream_supported(N, H, W, Cin, Cout, dt(x)))


def conv3x3_stream_geometry(N: int, H: int, W: int, Cin: int, Cout: int, dtype: torch.dtype) -> bool:
    """Whether dvt_conv3x3_stream takes this (frames, map, channel pair): what a model needs to know before it decides the
    channel padding of the layer (no tensors yet)."""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    return bool(L.load().dvt_conv3x3_stream_supported(N, H, W, Cin, Cout, _DT[dtype]))


def conv3x3_stream(x: Tensor, wp: Tensor, N: int, H: int, W: int, Cin: int, Cout: int, want_stats: bool = False,
                   residual: Optional[Tensor] = None):
    """3x3 / 1 / 1 convolution 64 -> 144 / 144 -> 64 (layer 1 of R(2+1)D-18) or 128 -> 288 / 288 -> 128 (layer 2) from LDS halo
    patches with streamed weights (dvt_conv3x3_stream); same contract as conv3x3_c64 (statistics with the 144- / 288-wide
    output, residual with the 64- / 128-wide one)."""
    _need_cuda(x, wp, residual)
    y = torch.empty((N * H * W, Cout), dtype=x.dtype, device=x.device)
    if residual is not None:
        assert residual.is_contiguous() and residual.dtype == x.dtype and residual.shape == y.shape
    lib = L.load()
    partial, parts = None, 0
    if want_stats:
        parts = int(lib.dvt_conv3x3_stream_stats_parts(N, H, W, Cin, Cout))
        partial = workspace((parts + 64) * 2 * Cout * 4, x.device, slot="bn_partial")
    nb = (x.numel() + y.numel() + wp.numel()) * x.element_size()
    with _timed(("conv", "halo3x3_stream", N * H * W, Cout, 9 * Cin, nb), 2.0 * N * H * W * Cout * 9 * Cin):
        L.check(lib.dvt_conv3x3_stream(x.data_ptr(), wp.data_ptr(), y.data_ptr(), _p(partial), _p(residual), N, H, W, Cin, Cout,
                                       dt(x), _stream()), "dvt_conv3x3_stream")
    return (y, partial, parts) if want_stats else y


def conv3x1_stream_supported(x: Tensor, wp: Tensor, N: int, T: int, HW: int, Cin: int, Cout: int) -> bool:
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or wp.dtype != x.dtype or (Cin, Cout) != (64, 144):
        return False
    if not (x.is_contiguous() and wp.is_contiguous() and tuple(wp.shape) == (Cout, 3 * Cin) and x.shape == (N * T * HW, Cin)):
        return False
    return bool(L.load().dvt_conv3x1_stream_supported(N, T, HW, Cin, Cout, dt(x)))


def conv3x1_stream(x: Tensor, wp: Tensor, N: int, T: int, HW: int, Cin: int, Cout: int) -> Tensor:
    """(3, 1) / 1 / (1, 0) convolution 64 -> 144 over the [T, HW] view of N clips (dvt_conv3x1_stream): the temporal half of
    R(2+1)D's layer-1 Conv2Plus1D as its data gradient."""
    _need_cuda(x, wp)
    y = torch.empty((N * T * HW, Cout), dtype=x.dtype, device=x.device)
    nb = (x.numel() + y.numel() + wp.numel()) * x.element_size()
    with _timed(("conv", "halo3x3_stream", N * T * HW, Cout, 3 * Cin, nb), 2.0 * N * T * HW * Cout * 3 * Cin):
        L.check(L.load().dvt_conv3x1_stream(x.data_ptr(), wp.data_ptr(), y.data_ptr(), N, T, HW, Cin, Cout, dt(x), _stream()),
                 "dvt_conv3x1_stream")
    return y


def conv3x1_stream_bn_bwd(dy: Tensor, wp: Tensor, z: Tensor, affine, N: int, T: int, HW: int, training: bool, *,
                          dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None, accumulate: bool = False):
    """Data gradient of the (3, 1) temporal convolution 144 -> 64 (as ``conv3x1_stream``) with the backward of the BatchNorm
    (+ ReLU) in front of that layer fused in (dvt_conv3x1_stream_bn_bwd): dy [N*T*HW, 64], z [N*T*HW, 144] the spatial half's
    output, affine = (mean, invstd, gamma, beta, c_valid, relu) of the BatchNorm that normalised it
    -> (dz [N*T*HW, 144], dgamma, dbeta).  The 144-plane data gradient is never stored (computed twice instead)."""
    _need_cuda(dy, wp, z)
    assert dy.dtype == z.dtype and z.shape == (N * T * HW, 144) and dy.shape == (N * T * HW, 64)
    aff, keep = _bn_affine(affine)
    if dgamma is None:
        assert not accumulate
        dgamma = torch.empty((144,), dtype=torch.float32, device=z.device)
        dbeta = torch.empty((144,), dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z)
    lib = L.load()
    ws = workspace(lib.dvt_conv3x1_stream_bn_bwd_workspace_bytes(N, T, HW), z.device)
    esz = z.element_size()
    nb = 2 * (dy.numel() + z.numel() + wp.numel()) * esz + dz.numel() * esz
    with _timed(("conv", "stream3x1_bn_bwd", N * T * HW, 144, 3 * 64, nb), 2 * 2.0 * N * T * HW * 144 * 3 * 64):
        L.check(lib.dvt_conv3x1_stream_bn_bwd(dy.data_ptr(), wp.data_ptr(), z.data_ptr(), C.byref(aff), dz.data_ptr(),
                                              dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), N, T, HW, int(training),
                                              int(accumulate), dt(z), _stream()), "dvt_conv3x1_stream_bn_bwd")
    return dz, dgamma, dbeta


def conv3x3_c64_wgrad_supported(x: Tensor, dz: Tensor, N: int, H: int, W: int, Cout: int = 64) -> bool:
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or dz.dtype != x.dtype:
        return False
    if Cout < 64 or Cout % 16:
        return False
    if not (x.is_contiguous() and dz.is_contiguous() and x.shape == (N * H * W, 64) and dz.shape == (N * H * W, Cout)):
        return False
    return bool(L.load().dvt_conv3x3_c64_wgrad_supported(N, H, W, dt(x)))


def conv3x3_c64_wgrad(x: Tensor, dz: Tensor, N: int, H: int, W: int, master: Tensor, *, accumulate: bool = False,
                      defer_reduce: bool = False, Cout: int = 64):
    """Weight gradient of the 64 -> 64 3x3 / 1 / 1 convolution from LDS halo patches (dvt_conv3x3_c64_wgrad), summed into
    ``master`` (the parameter's own gradient f32 [64, 64, 3, 3]; += when accumulate).  defer_reduce: -> pending, for
    ``splitk_reduce_pending`` / a carrying launch (the workgroups' partials live in the deferred-reduce scratch slot)."""
    _need_cuda(x, dz, master)
    assert master.dtype == torch.float32 and master.is_contiguous() and master.numel() == Cout * 64 * 9
    lib = L.load()
    pend = L.SplitKPending()
    nbytes = int(lib.dvt_conv3x3_c64_wgrad_workspace_bytes(N, H, W))
    ws = _deferred_workspace(nbytes, x.device, pend) if defer_reduce else workspace(nbytes, x.device, slot="conv3_wgrad")
    nb = (x.numel() + dz.numel()) * x.element_size() + master.numel() * 4
    with _timed(("conv", "halo3x3_c64_wgrad", 576, Cout, N * H * W, nb), 2.0 * N * H * W * Cout * 576):
        if Cout == 64:
            L.check(lib.dvt_conv3x3_c64_wgrad(x.data_ptr(), dz.data_ptr(), master.data_ptr(), ws.data_ptr(), N, H, W,
                                              int(accumulate), int(defer_reduce), C.byref(pend), dt(x), _stream()),
                    "dvt_conv3x3_c64_wgrad")
        else:       # (one launch + reduce per 64-channel group of dz over the same workspace; the last reduce deferred)
            L.check(lib.dvt_conv3x3_c64_wgrad_wide(x.data_ptr(), dz.data_ptr(), master.data_ptr(), ws.data_ptr(), N, H, W, Cout,
                                                   int(accumulate), int(defer_reduce), C.byref(pend), dt(x), _stream()),
                    "dvt_conv3x3_c64_wgrad_wide")
    if defer_reduce:
        pend._keep = (ws, master)
        return pend
    return None


def conv3x1_wgrad_supported(x: Tensor, dz: Tensor, N: int, T: int, Lp: int, Cin: int, Cout: int) -> bool:
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or dz.dtype != x.dtype:
        return False
    if not (x.is_contiguous() and dz.is_contiguous() and x.shape == (N * T * Lp, Cin) and dz.shape == (N * T * Lp, Cout)):
        return False
    return bool(L.load().dvt_conv3x1_wgrad_supported(N, T, Lp, Cin, Cout, dt(x)))


def _bn_affine(affine):
    """affine = (mean, invstd, gamma, beta, c_valid, relu) of the BatchNorm in front of a map, or None -> (struct, keep-alive)."""
    if affine is None:
        return None, None
    mean, invstd, gamma, beta, c_valid, relu = affine[:6]
    _need_cuda(mean, invstd, gamma, beta)
    for t in (mean, invstd, gamma, beta):
        assert t.dtype == torch.float32 and t.is_contiguous()
    a = L.BnAffine()
    a.mean, a.invstd, a.gamma, a.beta = mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    a.c_valid, a.relu = int(c_valid or 0), int(bool(relu))
    return a, (mean, invstd, gamma, beta)


def conv3x1_wgrad(x: Tensor, dz: Tensor, N: int, T: int, Lp: int, master: Tensor, *, accumulate: bool = False,
                  defer_reduce: bool = False, affine=None):
    """Weight gradient of the (3, 1) temporal convolution 144 -> 64 over the [T, H*W] view of N clips from LDS-resident
    sliding windows (dvt_conv3x1_wgrad), summed into ``master`` (f32 [64, 144, 3(, 1, 1)]; += when accumulate).
    affine: x is the convolution output in front of the BatchNorm (+ ReLU) whose (mean, invstd, gamma, beta, c_valid, relu)
    this is; the normalised activation is formed in the staged window.  defer_reduce: -> pending, as ``conv3x3_c64_wgrad``."""
    _need_cuda(x, dz, master)
    assert master.dtype == torch.float32 and master.is_contiguous() and master.numel() == 64 * 144 * 3
    lib = L.load()
    pend = L.SplitKPending()
    aff, keep = _bn_affine(affine)
    nbytes = int(lib.dvt_conv3x1_wgrad_workspace_bytes(N, T, Lp))
    ws = _deferred_workspace(nbytes, x.device, pend) if defer_reduce else workspace(nbytes, x.device, slot="conv3_wgrad")
    nb = (x.numel() + dz.numel()) * x.element_size() + master.numel() * 4
    with _timed(("conv", "window3x1_wgrad", 432, 64, N * T * Lp, nb), 2.0 * N * T * Lp * 64 * 432):
        L.check(lib.dvt_conv3x1_wgrad(x.data_ptr(), None if aff is None else C.byref(aff), dz.data_ptr(), master.data_ptr(),
                                      ws.data_ptr(), N, T, Lp, int(accumulate), int(defer_reduce), C.byref(pend), dt(x), _stream()),
                "dvt_conv3x1_wgrad")
    if defer_reduce:
        pend._keep = (ws, master, keep)
        return pend
    return None


def conv3x1_window_geometry(N: int, T: int, Lp: int, Cin: int, Cout: int, dtype: torch.dtype) -> bool:
    """Do BOTH window kernels of the (3, 1) temporal convolution (forward and weight gradient) take this geometry?  (No
    tensors needed: models/video_resnet.py asks before it decides to keep a BatchNorm virtual.)"""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    lib = L.load()
    return bool(lib.dvt_conv3x1_fwd_supported(N, T, Lp, Cin, Cout, _DT[dtype])) and \
        bool(lib.dvt_conv3x1_wgrad_supported(N, T, Lp, Cin, Cout, _DT[dtype]))


def conv3x1_fwd_supported(x: Tensor, wp: Tensor, N: int, T: int, Lp: int, Cin: int, Cout: int) -> bool:
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or wp.dtype != x.dtype:
        return False
    if not (x.is_contiguous() and wp.is_contiguous() and x.shape == (N * T * Lp, Cin) and wp.dim() == 2 and wp.shape[0] == Cout
            and wp.shape[1] >= 3 * Cin and wp.shape[1] % 8 == 0):
        return False
    return bool(L.load().dvt_conv3x1_fwd_supported(N, T, Lp, Cin, Cout, dt(x)))


def conv3x1_fwd(x: Tensor, wp: Tensor, N: int, T: int, Lp: int, want_stats: bool = False, affine=None):
    """(3, 1) temporal convolution 144 -> 64 (or 64 -> 64: the stem's temporal half / its data gradient) over the [T, H*W] view
    of N clips from LDS-resident sliding windows (dvt_conv3x1_fwd): x [N*T*Lp, Cin], wp [64, >= 3 * Cin] (column kt * Cin + ci)
    -> [N*T*Lp, 64]; want_stats: also the partial column sums for ``bn_stats_from_partials``; affine (144 channels only) as
    ``conv3x1_wgrad``."""
    _need_cuda(x, wp)
    lib = L.load()
    Cin = x.shape[1]
    y = torch.empty((N * T * Lp, 64), dtype=x.dtype, device=x.device)
    aff, _keep = _bn_affine(affine)
    partial, parts = None, 0
    if want_stats:
        parts = int(lib.dvt_conv3x1_fwd_stats_parts(N, T, Lp, Cin))
        partial = workspace((parts + 64) * 2 * 64 * 4, x.device, slot="bn_partial")
    nb = (x.numel() + wp.numel() + y.numel()) * x.element_size()
    with _timed(("conv", "window3x1_fwd" if Cin == 144 else "window3x1_c64", N * T * Lp, 64, 3 * Cin, nb), 2.0 * N * T * Lp * 64 * 3 * Cin):
        L.check(lib.dvt_conv3x1_fwd(x.data_ptr(), None if aff is None else C.byref(aff), wp.data_ptr(), wp.shape[1], y.data_ptr(),
                                    _p(partial), N, T, Lp, Cin, dt(x), _stream()), "dvt_conv3x1_fwd")
    return (y, partial, parts) if want_stats else y


def conv2d_implicit(x: Tensor, wp: Tensor, N: int, Cc: int, H: int, W: int, Cout: int, k, stride, pad,
                    want_stats: bool = False, trim_w: int = 0, carry=None, residual: Optional[Tensor] = None,
                    out: Optional[Tensor] = None, out_hw=None, out_rows: Optional[Tensor] = None, residual_compact: bool = False):
    """NHWC matrix x [N*H*W, C], packed weights wp [Cout, kh*kw*C] -> [N*Ho*Wo, Cout]; gather fused into the GEMM.
    want_stats: also returns (partial, parts), the per-block column sums / sums of squares of the output that the GEMM
    epilogue leaves for the BatchNorm behind the convolution (bn_stats_from_partials).
    out / out_hw / out_rows / residual_compact: one parity class of a strided convolution's data gradient
    (dvt_conv_desc.out_h / out_w / out_rows): the launch computes out_hw pixels per image and writes row m of its product
    to row out_rows[m] of ``out`` (a full-size map the classes fill between them)."""
    _need_cuda(x, wp)
    Ho, Wo = conv_out_hw(H, W, k, stride, pad)
    Wo -= trim_w                                  # columns dropped at the right edge (dvt_conv_desc.trim_w)
    if out_hw is not None:
        Ho, Wo = out_hw
    if out_rows is not None:
        assert out is not None and out.is_contiguous() and out.dtype == x.dtype and out.shape[1] == Cout and not want_stats
        assert out_rows.dtype == torch.int32 and out_rows.is_contiguous() and out_rows.numel() == N * Ho * Wo
        y = out
    else:
        assert out is None and not residual_compact
        y = torch.empty((N * Ho * Wo, Cout), dtype=x.dtype, device=x.device)
    d = _conv_desc(x, wp, y, N, Cc, H, W, Cout, k, stride, pad, trim_w)
    if out_hw is not None:
        d.out_h, d.out_w = Ho, Wo
    if out_rows is not None:
        d.out_rows, d.residual_compact = out_rows.data_ptr(), int(residual_compact)
    if residual is not None:                          # a second gradient path joining this one: added on the accumulators
        assert residual.is_contiguous() and residual.dtype == x.dtype and not want_stats
        assert residual.shape == ((N * Ho * Wo, Cout) if (residual_compact or out_rows is None) else y.shape)
        d.residual = residual.data_ptr()
    lib = L.load()
    partial, parts = None, 0
    if want_stats:
        parts = int(lib.dvt_conv2d_implicit_stats_parts(C.byref(d)))
        partial = workspace(lib.dvt_conv2d_implicit_stats_bytes(C.byref(d)), x.device, slot="bn_partial")
        d.stats_partial = _p(partial)
    (kh, kw) = _pair(k)
    rows = N * Ho * Wo
    nb = (x.numel() + wp.numel() + rows * Cout * (2 if residual is not None else 1)) * x.element_size()    # implicit GEMM: the image is read once, not kh*kw times
    if carry is not None and carry.valid:             # the layer's weight-gradient reduce rides in this launch's grid tail
        d.carry = C.addressof(carry)
    wsb = int(lib.dvt_conv2d_implicit_workspace_bytes(C.byref(d)))
    if wsb:                                           # few output rows x deep K: split-K slabs (dvt_conv2d_implicit_workspace_bytes)
        d.workspace = workspace(wsb, x.device, slot="conv_split").data_ptr()
    with _timed(("conv", "implicit", rows, Cout, kh * kw * Cc, nb), 2.0 * rows * Cout * kh * kw * Cc):
        L.check(lib.dvt_conv2d_implicit(C.byref(d), _stream()), "dvt_conv2d_implicit")
    if carry is not None:
        carry.valid = 0
    return (y, partial, parts) if want_stats else y


_CLASS_ROWS = {}


def strided_class_rows(N: int, H: int, W: int, sh: int, sw: int, a: int, b: int, device) -> Tensor:
    """int32 [N * Hq * Wq]: the row of the full-size NHWC map [N*H*W, C] that pixel (n, hq, wq) of parity class (a, b) --
    input pixel (sh * hq + a, sw * wq + b) -- names.  Static per geometry: built once on the host, cached on the device."""
    key = (N, H, W, sh, sw, a, b, str(device))
    t = _CLASS_ROWS.get(key)
    if t is None:
        import numpy as np
        hq = np.arange(a, H, sh, dtype=np.int64)
        wq = np.arange(b, W, sw, dtype=np.int64)
        rows = (np.arange(N, dtype=np.int64)[:, None, None] * (H * W) + hq[None, :, None] * W + wq[None, None, :]).reshape(-1)
        assert rows.size == 0 or rows.max() < (1 << 31)
        t = torch.from_numpy(rows.astype(np.int32)).to(device)
        _CLASS_ROWS[key] = t
    return t


def strided_dgrad_classes(k, stride, pad, H: int, W: int):
    """The parity classes of the data gradient of a (k, stride, pad) convolution over an H x W input, or None when the
    decomposition does not apply (a class without a tap -- e.g. a strided 1 x 1 -- or with taps above / left of the map).
    -> [(a, b, (nth, ntw), (ph', pw'), (rh, rw), (Hq, Wq))]: class (a, b) = input pixels (hi % sh, wi % sw); its gradient is the
    stride-1 convolution of dz with the nth x ntw taps ki = rh + j * sh (decreasing), padded by (ph', pw') in front."""
    (kh, kw), (sh, sw), (ph, pw) = _pair(k), _pair(stride), _pair(pad)
    if sh == 1 and sw == 1:
        return None
    out = []
    for a in range(sh):
        for b in range(sw):
            dims = []
            for (kk, s, p, r0, ext) in ((kh, sh, ph, a, H), (kw, sw, pw, b, W)):
                r = (r0 + p) % s
                nt = (kk - r + s - 1) // s if r < kk else 0
                padq = (nt - 1) - (r0 + p) // s
                q = (ext - r0 + s - 1) // s if ext > r0 else 0
                dims.append((nt, padq, r, q))
            (nth, pph, rh, Hq), (ntw, ppw, rw, Wq) = dims
            if Hq == 0 or Wq == 0:
                continue
            if nth == 0 or ntw == 0 or pph < 0 or ppw < 0:
                return None
            out.append((a, b, (nth, ntw), (pph, ppw), (rh, rw), (Hq, Wq)))
    return out


def bn_stats_from_partials(partial: Tensor, parts: int, rows: int, Cc: int, running_mean: Optional[Tensor],
                           running_var: Optional[Tensor], eps: float, momentum: float, c_valid: int = 0):
    """mean / invstd (and the running statistics update) from the partial sums of conv2d_implicit(want_stats=True).
    c_valid: channels the running statistics really have (channel-padded layers; 0 = Cc)."""
    mean = torch.empty((Cc,), dtype=torch.float32, device=partial.device)
    invstd = torch.empty((Cc,), dtype=torch.float32, device=partial.device)
    L.check(L.load().dvt_bn_stats_from_partials(partial.data_ptr(), parts, mean.data_ptr(), invstd.data_ptr(),
                                               _p(running_mean), _p(running_var), rows, Cc, c_valid, eps, momentum,
                                               _stream()), "dvt_bn_stats_from_partials")
    return mean, invstd


def conv2d_implicit_wgrad_supported(x: Tensor, dz: Tensor, N, Cc, H, W, Cout, k, stride, pad, trim_w: int = 0) -> bool:
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or dz.dtype != x.dtype:
        return False
    if not (x.is_contiguous() and dz.is_contiguous()):
        return False
    return bool(L.load().dvt_conv2d_implicit_wgrad_supported(C.byref(_conv_desc(x, dz, None, N, Cc, H, W, Cout, k, stride, pad, trim_w))))


def conv2d_implicit_wgrad(x: Tensor, dz: Tensor, N: int, Cc: int, H: int, W: int, Cout: int, k, stride, pad,
                          trim_w: int = 0, defer_reduce: bool = False, master: Optional[Tensor] = None,
                          accumulate: bool = False, logical: Optional[Tuple[int, int]] = None):
    """-> dWt f32 [kh*kw*C, Cout] = gather(x)^T dz, the column matrix never materialised.
    defer_reduce: -> (dWt, pending): the split-K reduce is left to the data-gradient launch of the same layer
    (``conv2d_implicit(..., carry=pending)`` / ``linear_dgrad(..., carry=pending)``) or ``splitk_reduce_pending``.
    master: the parameter's own gradient f32 [Cout, C, kh, kw] (+= when accumulate): the reduce scatters into it directly
    and it is what is returned in dWt's place (no packed intermediate, no scatter launch)."""
    _need_cuda(x, dz, master)
    (kh, kw) = _pair(k)
    cout_l, cin_l = logical if logical is not None else (Cout, Cc)       # channel-padded layers: the parameter's own counts
    if master is not None:
        assert master.dtype == torch.float32 and master.is_contiguous() and master.numel() == cout_l * cin_l * kh * kw
        assert cout_l <= Cout and cin_l <= Cc
        out = master
    else:
        assert not accumulate
        out = torch.empty((kh * kw * Cc, Cout), dtype=torch.float32, device=x.device)
    d = _conv_desc(x, dz, out, N, Cc, H, W, Cout, k, stride, pad, trim_w)
    d.wgrad_master_layout, d.wgrad_accumulate = int(master is not None), int(accumulate)
    d.wgrad_cout_l, d.wgrad_cin_l = (cout_l, cin_l) if master is not None else (0, 0)
    lib = L.load()
    pending = L.SplitKPending() if defer_reduce else None
    nws = lib.dvt_conv2d_implicit_wgrad_workspace_bytes(C.byref(d))
    ws = _deferred_workspace(nws, x.device, pending) if defer_reduce else workspace(nws, x.device)
    d.workspace = _p(ws)
    if defer_reduce:
        pending._keep = (out, ws)
        d.defer_reduce = 1
        d.pending = C.addressof(pending)
    rows = dz.shape[0]
    nb = (x.numel() + dz.numel()) * x.element_size() + out.numel() * out.element_size()
    with _timed(("conv", "wgrad", kh * kw * Cc, Cout, rows, nb), 2.0 * rows * Cout * kh * kw * Cc):
        L.check(lib.dvt_conv2d_implicit_wgrad(C.byref(d), _stream()), "dvt_conv2d_implicit_wgrad")
    return (out, pending) if defer_reduce else out


def conv_weight_unpack_grad_t(gt: Tensor, shape, *, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    Cout, Cin, kh, kw = shape
    if out is None:
        assert not accumulate
        out = torch.empty(tuple(shape), dtype=torch.float32, device=gt.device)
    L.check(L.load().dvt_conv_weight_unpack_grad_t(gt.data_ptr(), out.data_ptr(), Cout, Cin, kh, kw, int(accumulate),
                                                   _stream()), "dvt_conv_weight_unpack_grad_t")
    return out


def pad3_f32(src: Tensor, A: int, B: int, K: int, Ap: int, Bp: int) -> Tensor:
    """f32 [A, B, K] -> zero-extended [Ap, Bp, K]."""
    _need_cuda(src)
    src = src.detach().contiguous()
    assert src.dtype == torch.float32 and src.numel() == A * B * K
    dst = torch.empty((Ap, Bp, K), dtype=torch.float32, device=src.device)
    L.check(L.load().dvt_pad3_f32(src.data_ptr(), dst.data_ptr(), A, B, K, Ap, Bp, _stream()), "dvt_pad3_f32")
    return dst


def unpad3_f32(src: Tensor, A: int, B: int, K: int, Bp: int, *, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """slice [:A, :B, :] of f32 [Ap, Bp, K] (+= into ``out`` when accumulate)."""
    _need_cuda(src)
    assert src.dtype == torch.float32 and src.is_contiguous()
    if out is None:
        assert not accumulate
        out = torch.empty((A, B, K), dtype=torch.float32, device=src.device)
    L.check(L.load().dvt_unpad3_f32(src.data_ptr(), out.data_ptr(), A, B, K, Bp, int(accumulate), _stream()), "dvt_unpad3_f32")
    return out


def conv_weight_pack_group(entries) -> None:
    """Both packed forms of many convolution weights in one launch.  entries: (src f32 [cout_l, cin_l, kh*kw] contiguous,
    dst tensor, cout_l, cin_l, kh, kw, cout_p, cin_p, ld, kind[, (sh, sw, rh, rw)]) with kind 0 = forward operand [cout_p, ld],
    1 = data-gradient operand [cin_p, kh*kw*cout_p], 2 = the data-gradient operand of one parity class of a strided
    convolution [cin_p, nth*ntw*cout_p] (dvt_pack_entry.cls_*)."""
    if not entries:
        return
    arr = (L.PackEntry * len(entries))()
    for i, ent in enumerate(entries):
        (src, dst, cout_l, cin_l, kh, kw, cout_p, cin_p, ld, kind), cls = ent[:10], (ent[10] if len(ent) > 10 else None)
        _need_cuda(src, dst)
        assert src.dtype == torch.float32 and src.is_contiguous() and src.numel() == cout_l * cin_l * kh * kw
        e = arr[i]
        ntaps = kh * kw
        if kind == 2:                                  # one parity class of a strided convolution's data gradient
            e.cls_sh, e.cls_sw, e.cls_rh, e.cls_rw = cls
            ntaps = ((kh - cls[2] + cls[0] - 1) // cls[0]) * ((kw - cls[3] + cls[1] - 1) // cls[1])
        assert dst.is_contiguous() and dst.numel() == (cout_p * ld if kind == 0 else cin_p * ntaps * cout_p)
        e.src, e.dst = src.data_ptr(), dst.data_ptr()
        e.cout_l, e.cin_l, e.kh, e.kw, e.cout_p, e.cin_p, e.ld, e.kind, e.dtype = cout_l, cin_l, kh, kw, cout_p, cin_p, ld, kind, dt(dst)
    L.check(L.load().dvt_conv_weight_pack_group(C.cast(arr, C.c_void_p), len(entries), _stream()), "dvt_conv_weight_pack_group")


def conv_weight_pack(w: Tensor, ld: int, dtype: torch.dtype) -> Tensor:
    _need_cuda(w)
    w = w.detach().contiguous()
    assert w.dtype == torch.float32 and w.dim() == 4
    Cout, Cin, kh, kw = w.shape
    out = torch.empty((Cout, ld), dtype=dtype, device=w.device)
    L.check(L.load().dvt_conv_weight_pack(w.data_ptr(), out.data_ptr(), _DT[dtype], Cout, Cin, kh, kw, ld, _stream()),
            "dvt_conv_weight_pack")
    return out


def conv_weight_pack_dgrad(w: Tensor, dtype: torch.dtype) -> Tensor:
    """[Cout, Cin, kh, kw] f32 -> [Cin, kh*kw*Cout]: rotated taps, transposed channels (data-gradient operand)."""
    _need_cuda(w)
    w = w.detach().contiguous()
    Cout, Cin, kh, kw = w.shape
    out = torch.empty((Cin, kh * kw * Cout), dtype=dtype, device=w.device)
    L.check(L.load().dvt_conv_weight_pack_dgrad(w.data_ptr(), out.data_ptr(), _DT[dtype], Cout, Cin, kh, kw, _stream()),
            "dvt_conv_weight_pack_dgrad")
    return out


def conv_weight_unpack_grad(g: Tensor, shape, *, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    _need_cuda(g)
    Cout, Cin, kh, kw = shape
    if out is None:
        assert not accumulate
        out = torch.empty(tuple(shape), dtype=torch.float32, device=g.device)
    L.check(L.load().dvt_conv_weight_unpack_grad(g.data_ptr(), out.data_ptr(), Cout, Cin, kh, kw, g.shape[1],
                                                 int(accumulate), _stream()), "dvt_conv_weight_unpack_grad")
    return out


def bn_stats(z: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor], eps: float,
             momentum: float, c_valid: int = 0) -> Tuple[Tensor, Tensor]:
    _need_cuda(z)
    rows, Cc = z.shape
    mean = torch.empty((Cc,), dtype=torch.float32, device=z.device)
    invstd = torch.empty((Cc,), dtype=torch.float32, device=z.device)
    lib = L.load()
    ws = workspace(lib.dvt_bn_workspace_bytes(rows, Cc), z.device)
    with _timed(("hbm", "bn_stats", rows * Cc), z.numel() * z.element_size()):
        L.check(lib.dvt_bn_stats(z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _p(running_mean), _p(running_var),
                                 ws.data_ptr(), rows, Cc, c_valid, eps, momentum, dt(z), _stream()), "dvt_bn_stats")
    return mean, invstd


def bn_eval_invstd(running_var: Tensor, eps: float) -> Tensor:
    out = torch.empty_like(running_var)
    L.check(L.load().dvt_bn_eval_invstd(running_var.data_ptr(), out.data_ptr(), running_var.numel(), eps, _stream()),
            "dvt_bn_eval_invstd")
    return out


def bn_apply_fwd(z: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor, beta: Tensor, residual: Optional[Tensor],
                 relu: bool, want_mask: bool = False, c_valid: int = 0):
    """want_mask: -> (y, mask) with mask uint8 [rows, C/8], the ReLU mask bits for bn_bwd (layers with a residual branch)."""
    rows, Cc = z.shape
    y = torch.empty_like(z)
    mask = torch.empty((rows, Cc // 8), dtype=torch.uint8, device=z.device) if want_mask else None
    nb = z.numel() * z.element_size() * (2 + (residual is not None)) + (rows * (Cc // 8) if want_mask else 0)
    with _timed(("hbm", "bn_apply_fwd", rows * Cc), nb):
        L.check(L.load().dvt_bn_apply_fwd(z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                          beta.data_ptr(), _p(residual), y.data_ptr(), _p(mask), rows, Cc, c_valid, int(relu),
                                          dt(z), _stream()), "dvt_bn_apply_fwd")
    return (y, mask) if want_mask else y


def bn_bwd(dy: Tensor, z: Tensor, y: Optional[Tensor], mean: Tensor, invstd: Tensor, gamma: Tensor, relu: bool,
           training: bool, want_dres: bool, *, dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None,
           accumulate: bool = False, beta: Optional[Tensor] = None, mask: Optional[Tensor] = None, c_valid: int = 0):
    """ReLU mask: ``mask`` (bn_apply_fwd(want_mask=True)) if given, else ``y``, else recomputed from z (``beta`` given, no
    residual branch)."""
    rows, Cc = z.shape
    dz = torch.empty_like(z)
    dres = torch.empty_like(z) if want_dres else None
    if dgamma is None:
        assert not accumulate
        dgamma = torch.empty((c_valid or Cc,), dtype=torch.float32, device=z.device)
        dbeta = torch.empty((c_valid or Cc,), dtype=torch.float32, device=z.device)
    if mask is not None:
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == rows * (Cc // 8)
        y = None
    lib = L.load()
    ws = workspace(lib.dvt_bn_workspace_bytes(rows, Cc), z.device)
    # train-mode BatchNorm backward is two passes by construction (the sums over the batch, then the input gradient):
    # each reads dy and z (and the mask bytes, or y when the mask cannot be recomputed), the second writes dz (and the
    # shortcut's gradient)
    esz = z.element_size()
    per_pass = z.numel() * esz * (2 + (y is not None)) + (mask.numel() if mask is not None else 0)
    nb = 2 * per_pass + z.numel() * esz * (1 + int(want_dres))
    with _timed(("hbm", "bn_bwd", rows * Cc), nb):
        L.check(lib.dvt_bn_bwd(dy.data_ptr(), z.data_ptr(), _p(y), _p(mask), mean.data_ptr(), invstd.data_ptr(),
                               gamma.data_ptr(), _p(beta), dz.data_ptr(), _p(dres), dgamma.data_ptr(), dbeta.data_ptr(),
                               ws.data_ptr(), rows, Cc, c_valid, int(relu), int(training), int(accumulate), dt(z), _stream()),
                "dvt_bn_bwd")
    return dz, dres, dgamma, dbeta


def bn_relu_maxpool_fwd(z: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor, beta: Tensor, N: int, Cc: int, H: int, W: int,
                        relu: bool):
    """BatchNorm (+ ReLU) + max-pool(3, 2, 1) of the NHWC map z in one pass -> (pooled [N*Ho*Wo, C], argmax taps)."""
    Ho, Wo = conv_out_hw(H, W, 3, 2, 1)
    y = torch.empty((N * Ho * Wo, Cc), dtype=z.dtype, device=z.device)
    idx = torch.empty((N * Ho * Wo * Cc,), dtype=torch.uint8, device=z.device)
    L.check(L.load().dvt_bn_relu_maxpool_fwd(z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             y.data_ptr(), idx.data_ptr(), N, Cc, H, W, int(relu), dt(z), _stream()),
            "dvt_bn_relu_maxpool_fwd")
    return y, idx


def bn_bwd_pooled(dy_pool: Tensor, idx: Tensor, z: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor, beta: Tensor, N: int,
                  H: int, W: int, relu: bool, training: bool, *, dgamma: Optional[Tensor] = None,
                  dbeta: Optional[Tensor] = None, accumulate: bool = False):
    """bn_bwd with the incoming gradient gathered from the gradient of the max-pool(3, 2, 1) that follows the layer."""
    rows, Cc = z.shape
    assert rows == N * H * W
    dz = torch.empty_like(z)
    if dgamma is None:
        assert not accumulate
        dgamma = torch.empty((Cc,), dtype=torch.float32, device=z.device)
        dbeta = torch.empty((Cc,), dtype=torch.float32, device=z.device)
    lib = L.load()
    ws = workspace(lib.dvt_bn_workspace_bytes(rows, Cc), z.device)
    L.check(lib.dvt_bn_bwd_pooled(dy_pool.data_ptr(), idx.data_ptr(), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                  gamma.data_ptr(), beta.data_ptr(), dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                  ws.data_ptr(), N, Cc, H, W, int(relu), int(training), int(accumulate), dt(z), _stream()),
            "dvt_bn_bwd_pooled")
    return dz, dgamma, dbeta


def maxpool_fwd(x: Tensor, N: int, Cc: int, H: int, W: int, k: int, stride: int, pad: int):
    Ho, Wo = conv_out_hw(H, W, k, stride, pad)
    y = torch.empty((N * Ho * Wo, Cc), dtype=x.dtype, device=x.device)
    idx = torch.empty((N * Ho * Wo * Cc,), dtype=torch.uint8, device=x.device)
    L.check(L.load().dvt_maxpool_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, Cc, H, W, k, stride, pad, dt(x),
                                     _stream()), "dvt_maxpool_fwd")
    return y, idx


def maxpool_bwd(dy: Tensor, idx: Tensor, N: int, Cc: int, H: int, W: int, k: int, stride: int, pad: int) -> Tensor:
    dx = torch.empty((N * H * W, Cc), dtype=dy.dtype, device=dy.device)
    L.check(L.load().dvt_maxpool_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), N, Cc, H, W, k, stride, pad,
                                     dt(dy), _stream()), "dvt_maxpool_bwd")
    return dx


def transpose_last2(x: Tensor) -> Tensor:
    """[B, R, C] -> [B, C, R] contiguous."""
    _need_cuda(x)
    x = x.contiguous()
    B, R, Cc = x.shape
    out = torch.empty((B, Cc, R), dtype=x.dtype, device=x.device)
    L.check(L.load().dvt_transpose_last2(x.data_ptr(), out.data_ptr(), B, R, Cc, dt(x), _stream()),
            "dvt_transpose_last2")
    return out
