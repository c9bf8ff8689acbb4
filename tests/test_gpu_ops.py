"""Per-operator parity: HIP kernels (through the C ABI) vs the CPU oracle.

Tolerances: fp32 mode rel-L2 <= 1e-4 (north_star bound 1e-3; observed ~1e-6);
bf16 mode rel-L2 <= 1e-2 against the oracle evaluated on the same bf16-rounded
inputs (BASELINE.md section 2 protocol).  Integer/byte work (patchify gather,
casts) is bit-exact.
"""
import math

import numpy as np
import pytest
import torch

from oracle import clip_path as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

F32_TOL = 1e-4
BF16_TOL = 1e-2


@pytest.fixture(scope="module")
def dvt():
    import dvt_amd
    dvt_amd._lib.load()
    return dvt_amd


def _tol(dtype):
    return F32_TOL if dtype == torch.float32 else BF16_TOL


def _gelu_grad(x):
    """d/dx gelu_erf(x) = Phi(x) + x phi(x): what DVT_EPI_GELU leaves in ``aux`` and DVT_EPI_DGELU multiplies by."""
    x = x.float()
    return 0.5 * (1.0 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327


def _rnd(shape, dtype, gen, scale=1.0):
    """Random tensor already rounded to ``dtype``; returns (device tensor, cpu fp32 copy)."""
    x = (torch.randn(shape, generator=gen) * scale).to(dtype)
    return x.cuda(), x.float()


DTYPES = [torch.float32, torch.bfloat16, torch.float16]


# ------------------------------------------------------------------ elementwise
def test_cast_add_exact(dvt, device):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1003, generator=g)
    xb = dvt.ops.cast(x.cuda(), torch.bfloat16)
    assert torch.equal(xb.cpu(), x.to(torch.bfloat16))
    assert torch.equal(dvt.ops.cast(xb, torch.float32).cpu(), x.to(torch.bfloat16).float())
    a, b = torch.randn(4, 999, generator=g), torch.randn(4, 999, generator=g)
    assert torch.equal(dvt.ops.add(a.cuda(), b.cuda()).cpu(), a + b)


@pytest.mark.parametrize("P,H,W", [(16, 64, 64), (8, 32, 32), (4, 8, 12), (16, 224, 224)])
def test_patchify_bit_exact(dvt, device, P, H, W):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 3, H, W, generator=g)
    ref = O.patchify(x, P).reshape(-1, P * P * 3)
    out = dvt.ops.patchify(x.cuda(), P, torch.float32)
    assert torch.equal(out.cpu(), ref)
    outb = dvt.ops.patchify(x.cuda(), P, torch.bfloat16)
    assert torch.equal(outb.cpu(), ref.to(torch.bfloat16))
    back = dvt.ops.patchify_bwd(out, x.shape, P, torch.float32)
    assert torch.equal(back.cpu(), x)          # adjoint of a permutation = inverse


@pytest.mark.parametrize("dtype", DTYPES)
def test_tokens_assemble(dvt, device, dtype):
    g = torch.Generator().manual_seed(3)
    B, T, n, d = 2, 3, 5, 64
    S = B * T
    emb_d, emb = _rnd((S * n, d), dtype, g)
    cls = torch.randn(1, 1, d, generator=g)
    pos = torch.randn(1, T, n + 1, d, generator=g)
    embr, clsr, posr = (t.clone().requires_grad_(True) for t in (emb, cls, pos))
    ref = torch.cat((clsr.reshape(1, 1, d).expand(S, 1, d), embr.reshape(S, n, d)), 1) \
        + posr[0].repeat(B, 1, 1)
    e_d = emb_d.clone().requires_grad_(True)
    c_d = cls.cuda().requires_grad_(True)
    p_d = pos.cuda().requires_grad_(True)
    out = dvt.functional.tokens_assemble(e_d, c_d, p_d, S, T, n)
    assert rel_l2(out, ref) < _tol(dtype)
    gy_d, gy = _rnd(out.shape, dtype, g)
    ref.backward(gy)
    out.backward(gy_d)
    assert rel_l2(e_d.grad, embr.grad) < _tol(dtype)
    assert rel_l2(c_d.grad, clsr.grad) < 1e-5
    assert rel_l2(p_d.grad, posr.grad) < 1e-5


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,d", [(7, 64), (130, 128), (1030, 512), (28, 896), (9, 2048)])
def test_layernorm(dvt, device, dtype, rows, d):
    g = torch.Generator().manual_seed(4)
    x_d, x = _rnd((rows, d), dtype, g, 2.0)
    w = 1 + 0.2 * torch.randn(d, generator=g)
    b = 0.3 * torch.randn(d, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = O.layernorm(xr, wr, br)
    xd = x_d.clone().requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = dvt.functional.layernorm(xd, wd, bd)
    assert rel_l2(out, ref) < _tol(dtype)
    gy_d, gy = _rnd((rows, d), dtype, g)
    ref.backward(gy)
    out.backward(gy_d)
    assert rel_l2(xd.grad, xr.grad) < _tol(dtype)
    assert rel_l2(wd.grad, wr.grad) < 5e-5 if dtype == torch.float32 else rel_l2(wd.grad, wr.grad) < BF16_TOL
    assert rel_l2(bd.grad, br.grad) < 5e-5


def test_layernorm_strided_rows_and_fused_residual_grad(dvt, device):
    g = torch.Generator().manual_seed(5)
    S, N, d = 6, 5, 64
    x_d, x = _rnd((S, N, d), torch.float32, g)
    w, b = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    y, mean, rstd = dvt.ops.layernorm_fwd(x_d, w.cuda(), b.cuda(), 1e-5, rows=(S, 1, N * d, 0))
    assert rel_l2(y, O.layernorm(x[:, 0], w, b)) < F32_TOL
    # dx = LN'(dy) + dx_add in one kernel
    x2 = x.reshape(-1, d).clone().requires_grad_(True)
    ref = O.layernorm(x2, w, b)
    gy_d, gy = _rnd((S * N, d), torch.float32, g)
    add_d, add = _rnd((S * N, d), torch.float32, g)
    ref.backward(gy)
    yy, m2, r2 = dvt.ops.layernorm_fwd(x_d.view(-1, d), w.cuda(), b.cuda())
    dx, dg, db = dvt.ops.layernorm_bwd(gy_d, x_d.view(-1, d), w.cuda(), m2, r2, dx_add=add_d)
    assert rel_l2(dx, x2.grad + add) < F32_TOL


# ------------------------------------------------------------------ GEMM family
GEMM_SHAPES = [
    (64, 64, 64), (128, 128, 128), (200, 136, 72), (130, 264, 200), (1000, 256, 512),
    (8, 19, 512),        # 19-class head: generic path
    (136, 384, 128),     # C1-like token count (not a multiple of 64 rows)
    (257, 128, 1024),
]


@pytest.mark.parametrize("M,N", [(64, 256), (1024, 512), (1025, 512), (5000, 264), (264, 2048)])
def test_colsum_row_counts(dvt, device, M, N):
    """Both sides of the single-launch (<= 1024 rows) / two-pass switch, bf16 and f32, overwrite and accumulate."""
    g = torch.Generator().manual_seed(12)
    for dtype in (torch.bfloat16, torch.float32):
        x_d, x = _rnd((M, N), dtype, g)
        out = dvt.ops.colsum(x_d)
        assert rel_l2(out, x.sum(0)) < 1e-5
        dvt.ops.colsum(x_d, out=out, accumulate=True)
        assert rel_l2(out, 2 * x.sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_linear_fwd_dgrad_wgrad(dvt, device, dtype, M, N, K):
    g = torch.Generator().manual_seed(6)
    s = 1.0 / math.sqrt(K)
    x_d, x = _rnd((M, K), dtype, g)
    w_d, w = _rnd((N, K), dtype, g, s)
    bias = torch.randn(N, generator=g)
    y = dvt.ops.linear_fwd(x_d, w_d, bias.cuda())
    assert rel_l2(y, x @ w.t() + bias) < _tol(dtype)
    dy_d, dy = _rnd((M, N), dtype, g)
    dx = dvt.ops.linear_dgrad(dy_d, w_d)
    assert rel_l2(dx, dy @ w) < _tol(dtype)
    dw = dvt.ops.linear_wgrad(dy_d, x_d)
    assert dw.dtype == torch.float32
    assert rel_l2(dw, dy.t() @ x) < _tol(dtype)
    db = dvt.ops.colsum(dy_d)
    assert rel_l2(db, dy.sum(0)) < 1e-5
    dvt.ops.colsum(dy_d, out=db, accumulate=True)       # accumulate form (single-launch path for <= 1024 rows)
    assert rel_l2(db, 2 * dy.sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K", [192, 1024])      # 1024: split-K with the epilogue applied in the reduce
def test_gemm_epilogues(dvt, device, dtype, K):
    L = dvt._lib
    g = torch.Generator().manual_seed(7)
    M, N = 260, 136
    x_d, x = _rnd((M, K), dtype, g)
    w_d, w = _rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias = 0.5 * torch.randn(N, generator=g)
    res_d, res = _rnd((M, N), dtype, g)
    pre = x @ w.t() + bias
    tol = _tol(dtype)
    # GELU with its derivative saved for the backward epilogue
    aux = torch.empty((M, N), dtype=dtype, device="cuda")
    h = dvt.ops.linear_fwd(x_d, w_d, bias.cuda(), epilogue=L.EPI_GELU, aux=aux)
    assert rel_l2(aux, _gelu_grad(pre)) < tol
    assert rel_l2(h, O.gelu_erf(pre)) < tol
    # ReLU
    r = dvt.ops.linear_fwd(x_d, w_d, bias.cuda(), epilogue=L.EPI_RELU)
    assert rel_l2(r, torch.relu(pre)) < tol
    # residual
    y = dvt.ops.linear_fwd(x_d, w_d, bias.cuda(), epilogue=L.EPI_RESIDUAL, residual=res_d)
    assert rel_l2(y, pre + res) < tol
    # dgrad with GELU' / ReLU' epilogue: du = (dy @ W2) * act'(u)
    u_d, u = _rnd((M, K), dtype, g)
    dy_d, dy = _rnd((M, N), dtype, g)
    uu = u.clone().requires_grad_(True)
    O.gelu_erf(uu).backward(dy @ w)
    du = dvt.ops.linear_dgrad(dy_d, w_d, epilogue=L.EPI_DGELU, aux=dvt.ops.cast(_gelu_grad(u).cuda(), dtype))
    assert rel_l2(du, uu.grad) < tol
    hpos = torch.relu(u)
    dr = dvt.ops.linear_dgrad(dy_d, w_d, epilogue=L.EPI_DRELU, aux=dvt.ops.cast(hpos.cuda(), dtype))
    assert rel_l2(dr, (dy @ w) * (hpos > 0)) < tol


def test_gemm256_streaming_store_path(dvt, device):
    """The FF1 shape of the metric workload: its two outputs (414 MB) exceed the streaming threshold, so C and the saved
    derivative are written with non-temporal stores; reference = fp32 matmul on the GPU."""
    L = dvt._lib
    g = torch.Generator().manual_seed(5)
    M, N, K = 50432, 2048, 512
    x = (torch.randn(M, K, generator=g) * 1.0).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).cuda()
    bias = (0.5 * torch.randn(N, generator=g)).cuda()
    aux = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    y = dvt.ops.linear_fwd(x, w, bias, epilogue=L.EPI_GELU, aux=aux)
    pre = x.float() @ w.float().t() + bias
    gp = _gelu_grad(pre)
    assert float((aux.float() - gp).norm() / gp.norm()) < BF16_TOL
    ref = torch.nn.functional.gelu(pre)
    assert float((y.float() - ref).norm() / ref.norm()) < BF16_TOL


@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16])
def test_gemm256_lds_dma_kernel_all_layouts_and_epilogues(dvt, device, dt16):
    """Shapes large enough for the 256x256 LDS-DMA kernel (>= 96 tiles), with ragged
    M (6208 = 24.25 tiles) and N edges, all three operand layouts, split-K and every
    fused epilogue."""
    L = dvt._lib
    g = torch.Generator().manual_seed(21)
    M, N, K = 6208, 1024, 1024
    x_d, x = _rnd((M, K), dt16, g)
    w_d, w = _rnd((N, K), dt16, g, 1 / math.sqrt(K))
    bias = 0.5 * torch.randn(N, generator=g)
    pre = x @ w.t() + bias
    y = dvt.ops.linear_fwd(x_d, w_d, bias.cuda())
    assert rel_l2(y, pre) < BF16_TOL
    aux = torch.empty((M, N), dtype=dt16, device="cuda")
    h = dvt.ops.linear_fwd(x_d, w_d, bias.cuda(), epilogue=L.EPI_GELU, aux=aux)
    assert rel_l2(aux, _gelu_grad(pre)) < BF16_TOL and rel_l2(h, O.gelu_erf(pre)) < BF16_TOL
    res_d, res = _rnd((M, N), dt16, g)
    yr = dvt.ops.linear_fwd(x_d, w_d, bias.cuda(), epilogue=L.EPI_RESIDUAL, residual=res_d)
    assert rel_l2(yr, pre + res) < BF16_TOL
    dy_d, dy = _rnd((M, N), dt16, g)
    dx = dvt.ops.linear_dgrad(dy_d, w_d)                      # k-major x mn-major
    assert rel_l2(dx, dy @ w) < BF16_TOL
    u_d, u = _rnd((M, K), dt16, g)
    uu = u.clone().requires_grad_(True)
    O.gelu_erf(uu).backward(dy @ w)
    du = dvt.ops.linear_dgrad(dy_d, w_d, epilogue=L.EPI_DGELU, aux=dvt.ops.cast(_gelu_grad(u).cuda(), dt16))
    assert rel_l2(du, uu.grad) < BF16_TOL
    dw = dvt.ops.linear_wgrad(dy_d, x_d)                      # mn-major x mn-major, split-K
    assert rel_l2(dw, dy.t() @ x) < BF16_TOL
    assert torch.equal(dw, dvt.ops.linear_wgrad(dy_d, x_d))
    # bias gradient fused into the same pass (all-ones MFMA), overwrite and accumulate
    db = torch.full((N,), 7.0, device="cuda")
    dw2 = dvt.ops.linear_wgrad(dy_d, x_d, bias_out=db)
    assert torch.equal(dw2, dw) and rel_l2(db, dy.sum(0)) < 1e-5
    dvt.ops.linear_wgrad(dy_d, x_d, bias_out=db, bias_accumulate=True)
    assert rel_l2(db, 2 * dy.sum(0)) < 1e-5
    # ragged N (1000 = 3.9 tiles) and a narrower K
    w2_d, w2 = _rnd((1000, 192), dt16, g, 0.1)
    x2_d, x2 = _rnd((M, 192), dt16, g)
    y2 = dvt.ops.linear_fwd(x2_d, w2_d)
    assert rel_l2(y2, x2 @ w2.t()) < BF16_TOL


@pytest.mark.parametrize("M,N,K", [(50432, 512, 512), (50432, 2048, 512), (12345 * 8, 1536, 256), (23000, 1000, 192),
                                   (50432, 512, 2048), (3000 * 8, 256, 64),
                                   (50395, 512, 512)])     # 224-row tiles (configuration 8) with a ragged last tile
def test_gemm_large_m_every_row_written_exactly_once(dvt, device, M, N, K):
    """The metric workload's token counts (M = 50,432 and ragged variants) through the LDS-DMA kernel's XCD-aware tile
    remap: every output row must be written exactly once -- ragged M (not a multiple of the tile), ragged N, 1 / 2 / 4 / 6
    / 8 column panels, forward and data-gradient layouts, every fused epilogue.  The outputs start as NaN and are checked
    per element (max error), not only in norm: one misplaced 64-row block is 0.1 % of the rows."""
    L = dvt._lib
    g = torch.Generator().manual_seed(M + N + K)
    x_d = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w_d = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).cuda()
    bias = (0.5 * torch.randn(N, generator=g)).cuda()
    pre = x_d.float() @ w_d.float().t() + bias                     # fp32 reference on the same rounded operands
    scale = float(pre.abs().max())

    def check(got, want, what):
        err = float((got.float() - want).abs().max())
        assert err < 2e-2 * max(scale, float(want.abs().max())), (what, err)
        assert rel_l2(got, want) < BF16_TOL, what

    poison = lambda *shape: torch.full(shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    y = dvt.ops.gemm(x_d, w_d, M, N, K, a_kmajor=True, b_kmajor=True, lda=K, ldb=K, bias=bias, out=poison(M, N))
    check(y, pre, "plain")                                          # NaN anywhere = a row nobody wrote
    aux = poison(M, N)
    h = dvt.ops.gemm(x_d, w_d, M, N, K, a_kmajor=True, b_kmajor=True, lda=K, ldb=K, bias=bias, epilogue=L.EPI_GELU, aux=aux,
                     out=poison(M, N))
    check(aux, _gelu_grad(pre), "gelu derivative")
    check(h, torch.nn.functional.gelu(pre), "gelu")
    res_d = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    yr = dvt.ops.gemm(x_d, w_d, M, N, K, a_kmajor=True, b_kmajor=True, lda=K, ldb=K, bias=bias, epilogue=L.EPI_RESIDUAL,
                      residual=res_d, out=poison(M, N))
    check(yr, pre + res_d.float(), "residual")
    # data gradient layout: dx[M, N] = dy[M, K] @ W[K, N] (mn-major B), plain and with GELU'
    wt_d = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(torch.bfloat16).cuda()
    dx_ref = x_d.float() @ wt_d.float()
    dx = dvt.ops.gemm(x_d, wt_d, M, N, K, a_kmajor=True, b_kmajor=False, lda=K, ldb=N, out=poison(M, N))
    check(dx, dx_ref, "dgrad")
    u_d = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    uu = u_d.float().requires_grad_(True)
    torch.nn.functional.gelu(uu).backward(dx_ref)
    du = dvt.ops.gemm(x_d, wt_d, M, N, K, a_kmajor=True, b_kmajor=False, lda=K, ldb=N, epilogue=L.EPI_DGELU,
                      aux=_gelu_grad(u_d).to(torch.bfloat16), out=poison(M, N))
    check(du, uu.grad, "dgrad + gelu'")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(264, 1536, 512), (264, 512, 2048), (264, 2048, 512), (8, 512, 512), (33, 64, 200),
                                   (528, 1000, 520)])
def test_gemm_launch_bound_shapes_panel_streaming_kernel(dvt, device, dtype, M, N, K):
    """The 33-token temporal encoder at B = 8 (264 rows; vit.py:122-128) and the heads: gemm_small.hip streams whole
    operand panels through LDS (no split-K, one launch per Linear).  Forward with every epilogue, data gradient with
    GELU', weight gradient with the fused bias gradient (overwrite and accumulate) and fp32 accumulation -- per element."""
    L = dvt._lib
    g = torch.Generator().manual_seed(M * 7 + N + K)
    tol = _tol(dtype)
    x_d, x = _rnd((M, K), dtype, g)
    w_d, w = _rnd((N, K), dtype, g, 1 / math.sqrt(K))
    bias = 0.5 * torch.randn(N, generator=g)
    pre = x @ w.t() + bias

    def check(got, want, what):
        assert torch.isfinite(got.float()).all(), what
        assert float((got.float().cpu() - want).abs().max()) < 4 * tol * float(want.abs().max()), what
        assert rel_l2(got, want) < tol, what

    nan16 = lambda *shape: torch.full(shape, float("nan"), dtype=dtype, device="cuda")
    check(dvt.ops.gemm(x_d, w_d, M, N, K, a_kmajor=True, b_kmajor=True, lda=K, ldb=K, bias=bias.cuda(), out=nan16(M, N)), pre,
          "plain")
    aux = nan16(M, N)
    h = dvt.ops.gemm(x_d, w_d, M, N, K, a_kmajor=True, b_kmajor=True, lda=K, ldb=K, bias=bias.cuda(), epilogue=L.EPI_GELU,
                     aux=aux, out=nan16(M, N))
    check(aux, _gelu_grad(pre), "gelu derivative")
    check(h, O.gelu_erf(pre), "gelu")
    res_d, res = _rnd((M, N), dtype, g)
    check(dvt.ops.linear_fwd(x_d, w_d, bias.cuda(), epilogue=L.EPI_RESIDUAL, residual=res_d), pre + res, "residual")
    check(dvt.ops.linear_fwd(x_d, w_d, bias.cuda(), epilogue=L.EPI_RELU), torch.relu(pre), "relu")
    dy_d, dy = _rnd((M, N), dtype, g)
    check(dvt.ops.linear_dgrad(dy_d, w_d), dy @ w, "dgrad")
    u_d, u = _rnd((M, K), dtype, g)
    uu = u.clone().requires_grad_(True)
    O.gelu_erf(uu).backward(dy @ w)
    check(dvt.ops.linear_dgrad(dy_d, w_d, epilogue=L.EPI_DGELU, aux=dvt.ops.cast(_gelu_grad(u).cuda(), dtype)), uu.grad,
          "dgrad gelu'")
    dw_ref = dy.t() @ x
    db = torch.full((N,), float("nan"), device="cuda")
    dw = dvt.ops.linear_wgrad(dy_d, x_d, bias_out=db)
    check(dw, dw_ref, "wgrad")
    assert rel_l2(db, dy.sum(0)) < 1e-5
    dvt.ops.linear_wgrad(dy_d, x_d, out=dw, accumulate=True, bias_out=db, bias_accumulate=True)
    check(dw, 2 * dw_ref, "wgrad accumulate")
    assert rel_l2(db, 2 * dy.sum(0)) < 1e-5
    assert torch.equal(dvt.ops.linear_wgrad(dy_d, x_d), dvt.ops.linear_wgrad(dy_d, x_d))      # no atomics: reproducible


@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,epi", [(264, 1536, 512, "none"), (264, 512, 2048, "dgelu"), (256, 512, 512, "none"),
                                       (8, 512, 512, "none"), (33, 192, 64, "drelu"), (50432 // 8, 512, 512, "none")])
def test_linear_backward_pair_equals_the_two_launches(dvt, device, dt16, M, N, K, epi):
    """ops.linear_backward: weight + data gradient of one Linear.  For launch-bound shapes both run in ONE launch
    (dvt_gemm_pair, the same tile code): bit-identical to the two separate calls, incl. the fused bias gradient, the
    accumulate flags and the activation-derivative epilogues; a full-size shape takes the deferred-reduce path."""
    g = torch.Generator().manual_seed(31)
    dy, _ = _rnd((M, N), dt16, g)
    x, _ = _rnd((M, K), dt16, g)
    w, _ = _rnd((N, K), dt16, g, 1 / math.sqrt(K))
    aux, _ = _rnd((M, K), dt16, g)
    L = dvt._lib
    e = {"none": L.EPI_NONE, "dgelu": L.EPI_DGELU, "drelu": L.EPI_DRELU}[epi]
    a = aux if epi != "none" else None
    dw0 = torch.full((N, K), 0.25, device="cuda")
    db0 = torch.full((N,), -1.0, device="cuda")
    dw_ref = dvt.ops.linear_wgrad(dy, x, out=dw0.clone(), accumulate=True, bias_out=db0.clone(), bias_accumulate=True)
    db_ref = db0 + dy.float().sum(0)
    dx_ref = dvt.ops.linear_dgrad(dy, w, epilogue=e, aux=a)
    dw, db = dw0.clone(), db0.clone()
    dw_out, dx = dvt.ops.linear_backward(dy, x, w, out=dw, accumulate=True, bias_out=db, bias_accumulate=True, epilogue=e, aux=a)
    assert dw_out.data_ptr() == dw.data_ptr()
    small = M <= 512
    if small:
        assert torch.equal(dw, dw_ref) and torch.equal(dx, dx_ref)
    else:
        assert rel_l2(dw, dw_ref) < 1e-6 and torch.equal(dx, dx_ref)
    assert rel_l2(db, db_ref) < 1e-5
    # fresh outputs, no bias
    dw2, dx2 = dvt.ops.linear_backward(dy, x, w, epilogue=e, aux=a)
    assert rel_l2(dw2, dw_ref - 0.25) < 1e-4 and torch.equal(dx2, dx_ref)


@pytest.mark.parametrize("rows", [28, 2, 13, 100])
def test_wgrad_ragged_row_count_runs_on_mfma(dvt, device, rows):
    """Weight gradients of the 14-token encoders (K = B * 14 = 28 rows, frame_transformer.py:204) -- K is not a multiple of 8
    but both operands are mn-major, so the MFMA kernel (with its zero-filled K tail) serves them."""
    g = torch.Generator().manual_seed(5)
    dy_d, dy = _rnd((rows, 896), torch.bfloat16, g)
    x_d, x = _rnd((rows, 512), torch.bfloat16, g)
    dw = dvt.ops.linear_wgrad(dy_d, x_d)
    assert dw.shape == (896, 512) and rel_l2(dw, dy.t() @ x) < 1e-5          # fp32 accumulation of bf16 products
    bias = torch.zeros(896, device="cuda")
    dvt.ops.linear_wgrad(dy_d, x_d, bias_out=bias)
    assert rel_l2(bias, dy.sum(0)) < 1e-5


def test_wgrad_split_k_reproducible(dvt, device):
    """Token-count reduction (K = rows) with few output tiles -> split-K slabs."""
    g = torch.Generator().manual_seed(8)
    M, N, K = 8192, 128, 256          # dW[N,K] = dy[M,N]^T x[M,K]
    dy_d, dy = _rnd((M, N), torch.bfloat16, g)
    x_d, x = _rnd((M, K), torch.bfloat16, g)
    a = dvt.ops.linear_wgrad(dy_d, x_d)
    b = dvt.ops.linear_wgrad(dy_d, x_d)
    assert torch.equal(a, b)                       # fixed summation order
    assert rel_l2(a, dy.t() @ x) < BF16_TOL
    acc = torch.ones((N, K), device="cuda")
    dvt.ops.linear_wgrad(dy_d, x_d, out=acc, accumulate=True)
    assert rel_l2(acc, dy.t() @ x + 1) < BF16_TOL


@pytest.mark.parametrize("shape", [(50432, 2048, 512), (50432, 512, 2048), (8192, 1536, 512), (20000, 512, 512)])
def test_deferred_split_k_reduce_rides_in_the_data_gradient_launch(dvt, device, shape):
    """A Linear's backward (vit.py:20-25,39-43): dW = dy^T x with its split-K reduce left undone (defer_reduce) and
    handed to the data-gradient GEMM dx = dy W behind it (carry), which performs it in tail workgroups of its own
    grid.  Results must be BIT-identical to the two-launch form (same slice order), for dW, the fused bias gradient
    and dx; accumulate into a destination; a pending reduce can also be flushed stand-alone or carried by a launch that
    is not the LDS-DMA kernel (then it runs first, as a launch of its own)."""
    g = torch.Generator().manual_seed(33)
    M, N, K = shape                              # dy [M, N], x [M, K], W [N, K]
    dy_d, _ = _rnd((M, N), torch.bfloat16, g)
    x_d, _ = _rnd((M, K), torch.bfloat16, g)
    w_d, _ = _rnd((N, K), torch.bfloat16, g, K ** -0.5)
    ops = dvt.ops
    dw_ref = torch.zeros((N, K), device="cuda")
    db_ref = torch.zeros((N,), device="cuda")
    ops.linear_wgrad(dy_d, x_d, out=dw_ref, bias_out=db_ref)
    dx_ref = ops.linear_dgrad(dy_d, w_d)
    # deferred + carried
    dw = torch.full((N, K), float("nan"), device="cuda")
    db = torch.full((N,), float("nan"), device="cuda")
    _, pend = ops.linear_wgrad(dy_d, x_d, out=dw, bias_out=db, defer_reduce=True)
    dx = ops.linear_dgrad(dy_d, w_d, carry=pend)
    assert torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    # accumulate into existing gradients
    dw2 = torch.ones((N, K), device="cuda")
    db2 = torch.ones((N,), device="cuda")
    _, pend = ops.linear_wgrad(dy_d, x_d, out=dw2, accumulate=True, bias_out=db2, bias_accumulate=True, defer_reduce=True)
    ops.linear_dgrad(dy_d, w_d, carry=pend)
    assert rel_l2(dw2, dw_ref + 1) < 1e-6 and rel_l2(db2, db_ref + 1) < 1e-6
    # stand-alone flush
    dw3 = torch.full((N, K), float("nan"), device="cuda")
    _, pend = ops.linear_wgrad(dy_d, x_d, out=dw3, defer_reduce=True)
    ops.splitk_reduce_pending(pend)
    assert torch.equal(dw3, dw_ref)
    # carried by a launch of the panel-streaming kernel (264 rows): performed first, as a launch of its own
    dw4 = torch.full((N, K), float("nan"), device="cuda")
    _, pend = ops.linear_wgrad(dy_d, x_d, out=dw4, defer_reduce=True)
    small = ops.linear_dgrad(dy_d[:264].contiguous(), w_d, carry=pend)
    assert torch.equal(dw4, dw_ref) and rel_l2(small, dx_ref[:264]) < 1e-2      # (another kernel: another summation order)


def test_second_deferred_reduce_while_one_is_pending_is_refused(dvt, device):
    """The slabs of a deferred split-K reduce sit in ONE scratch slot per stream: deferring a second reduce
    while the first is still pending would overwrite them, so it must raise instead of yielding a wrong dW; once the first
    is resolved (carried or flushed) deferring works again."""
    g = torch.Generator().manual_seed(34)
    M, N, K = 50432, 512, 512
    dy_d, _ = _rnd((M, N), torch.bfloat16, g)
    x_d, _ = _rnd((M, K), torch.bfloat16, g)
    ops = dvt.ops
    ref = ops.linear_wgrad(dy_d, x_d)
    dw = torch.full((N, K), float("nan"), device="cuda")
    _, pend = ops.linear_wgrad(dy_d, x_d, out=dw, defer_reduce=True)
    assert pend.valid                                           # this shape does split K
    with pytest.raises(RuntimeError, match="still pending"):
        ops.linear_wgrad(dy_d, x_d, defer_reduce=True)
    ops.splitk_reduce_pending(pend)                             # the first one's slabs were not touched by the refusal
    assert torch.equal(dw, ref)
    dw2 = torch.full((N, K), float("nan"), device="cuda")
    _, pend2 = ops.linear_wgrad(dy_d, x_d, out=dw2, defer_reduce=True)
    ops.splitk_reduce_pending(pend2)
    assert torch.equal(dw2, ref)


def test_wgrad_deep_k_fused_bias_with_exact_workspace(dvt, device):
    """Reference-default ViViT (dim 192) at batch 32: the to_out weight gradient is M = N = 192 with K = 100,864 token
    rows -> one 256x256 tile, 197 K slices, bias gradient fused (one scratch row of M floats per slice).  A C-ABI caller
    allocates exactly ``dvt_gemm_workspace_bytes``: the scratch must fit (it was sized for 128 slices), checked with a
    guard region behind the workspace."""
    import ctypes as C
    from dvt_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(18)
    rows, N, K = 100864, 192, 192
    dy_d, dy = _rnd((rows, N), torch.bfloat16, g)
    x_d, x = _rnd((rows, K), torch.bfloat16, g)
    dw = torch.empty((N, K), dtype=torch.float32, device="cuda")
    db = torch.empty((N,), dtype=torch.float32, device="cuda")
    d = L.GemmDesc()
    d.A, d.B, d.C = dy_d.data_ptr(), x_d.data_ptr(), dw.data_ptr()
    d.M, d.N, d.K = N, K, rows
    d.lda, d.ldb, d.ldc = N, K, K
    d.a_kmajor, d.b_kmajor = 0, 0
    d.in_dtype, d.out_dtype = L.BF16, L.F32
    d.epilogue, d.accumulate, d.alpha, d.split_k = L.EPI_NONE, 0, 1.0, 0
    d.colsum_out, d.colsum_accumulate = db.data_ptr(), 0
    need = lib.dvt_gemm_workspace_bytes(C.byref(d))
    guard = 1 << 20
    buf = torch.full((need + guard,), 0x5A, dtype=torch.uint8, device="cuda")
    d.workspace = buf.data_ptr()
    L.check(lib.dvt_gemm(C.byref(d), dvt.ops._stream()), "dvt_gemm")
    torch.cuda.synchronize()
    assert bool((buf[need:] == 0x5A).all()), "dvt_gemm wrote past the workspace size it reported"
    assert rel_l2(dw, dy.t() @ x) < BF16_TOL and rel_l2(db, dy.sum(0)) < 1e-5
    assert need >= 197 * N * 4                      # one scratch row per K slice actually planned


def test_gemm_rejects_bad_arguments(dvt, device):
    x = torch.randn(4, 8, device="cuda")
    w = torch.randn(6, 8, device="cuda")
    with pytest.raises(RuntimeError, match="RESIDUAL"):
        dvt.ops.linear_fwd(x, w, None, epilogue=dvt._lib.EPI_RESIDUAL)
    with pytest.raises(RuntimeError, match="GPU"):
        dvt.ops.linear_fwd(x.cpu(), w.cpu())


# ------------------------------------------------------------------ attention
ATTN_CASES = [
    # B, H, Lq, Lk, dh
    (2, 2, 17, 17, 64), (2, 3, 33, 33, 64), (1, 2, 197, 197, 64), (2, 2, 5, 40, 64),
    (1, 1, 64, 64, 64), (2, 2, 14, 14, 32), (1, 2, 15, 14, 448), (1, 4, 14, 15, 224),
    # short sequences of any head width (one workgroup per (b, h), csrc/attention.hip: attn_small_*): the frametransformer's
    # 15-token encoder at 8 heads, widths that are no multiple of 8 / of 4, the largest shape the kernels take, one query
    (2, 8, 15, 15, 112), (1, 3, 9, 11, 100), (1, 2, 7, 5, 30), (1, 2, 32, 32, 128), (2, 1, 1, 7, 20), (1, 1, 32, 32, 512),
    (2, 2, 325, 325, 64),          # long-clip tokens per frame (288^2): > 80 KiB of LDS, one 11-wave workgroup per CU
    (1, 2, 40, 600, 64),
    # every compile-time key / query count of the register-resident forward (<= 224 keys) and the unrolled backward
    # (<= 256), incl. Lq != Lk and the 225..256 band (online-softmax forward, unrolled backward)
    (1, 2, 90, 90, 64), (1, 2, 100, 128, 64), (1, 1, 150, 150, 64), (1, 2, 180, 190, 64), (1, 1, 250, 250, 64),
    (1, 1, 224, 224, 64),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Lq,Lk,dh", ATTN_CASES)
def test_attention_core(dvt, device, dtype, B, H, Lq, Lk, dh):
    g = torch.Generator().manual_seed(9)
    q_d, q = _rnd((B, H, Lq, dh), dtype, g)
    k_d, k = _rnd((B, H, Lk, dh), dtype, g)
    v_d, v = _rnd((B, H, Lk, dh), dtype, g)
    scale = dh ** -0.5
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = O.attention_core(qr, kr, vr, scale)
    qd, kd, vd = (t.clone().requires_grad_(True) for t in (q_d, k_d, v_d))
    out = dvt.functional.attention_core(qd, kd, vd, scale)
    tol = _tol(dtype)
    assert rel_l2(out, ref) < tol
    go_d, go = _rnd((B, H, Lq, dh), dtype, g)
    ref.backward(go)
    out.backward(go_d)
    assert rel_l2(qd.grad, qr.grad) < 2 * tol
    assert rel_l2(kd.grad, kr.grad) < 2 * tol
    assert rel_l2(vd.grad, vr.grad) < 2 * tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,Lq,Lk", [(3, 8, 197, 197), (2, 8, 33, 33), (1, 2, 193, 224), (1, 1, 16, 5), (2, 2, 96, 70)])
def test_attention_backward_one_pass_matches_the_kernel_pair(dvt, device, dtype, B, H, Lq, Lk):
    """vit.py:51-55 backward.  Sequences whose lengths pad to the same multiple of 32 (<= 224) run the backward as ONE pass
    (Q, dO, K staged once; dK / dV in registers of key-owner waves; dQ by a dedicated wave from 16-bit dS strips in LDS);
    the dq + dk/dv kernel pair stays reachable through dvt_attn_desc.bwd_two_pass.  Both against float64 on the same 16-bit
    operands, on the packed [tokens, 3 h dh] layout the blocks use (strided q / k / v views), and the one-pass form twice:
    bitwise reproducible (fixed summation order, no atomics)."""
    ops = dvt.ops
    g = torch.Generator().manual_seed(77)
    dh = 64
    L = max(Lq, Lk)
    qkv = (torch.randn(B, L, 3, H, dh, generator=g) * 0.7).to(dtype).cuda()
    q = qkv[:, :Lq, 0].permute(0, 2, 1, 3)
    k = qkv[:, :Lk, 1].permute(0, 2, 1, 3)
    v = qkv[:, :Lk, 2].permute(0, 2, 1, 3)
    o_mem = torch.empty(B, Lq, H, dh, dtype=dtype, device="cuda")
    o = o_mem.permute(0, 2, 1, 3)
    scale = dh ** -0.5
    lse = ops.attention_fwd(q, k, v, o, scale)
    do_mem = torch.randn(B, Lq, H, dh, generator=g).to(dtype).cuda()
    do = do_mem.permute(0, 2, 1, 3)

    def run(two_pass):
        dqkv = torch.full_like(qkv, float("nan"))
        dq = dqkv[:, :Lq, 0].permute(0, 2, 1, 3)
        dk = dqkv[:, :Lk, 1].permute(0, 2, 1, 3)
        dv = dqkv[:, :Lk, 2].permute(0, 2, 1, 3)
        ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, scale, two_pass=two_pass)
        return dq.clone(), dk.clone(), dv.clone()

    one, again, pair = run(False), run(False), run(True)
    for a, b in zip(one, again):
        assert torch.equal(a, b)
    q64, k64, v64, do64 = (t.double().cpu().requires_grad_(t is not do) for t in (q, k, v, do))
    p64 = torch.softmax(q64 @ k64.transpose(-1, -2) * scale, -1)
    (p64 @ v64).backward(do64)
    tol = 1.2e-2 if dtype == torch.bfloat16 else 2e-3
    for got, got2, ref in zip(one, pair, (q64.grad, k64.grad, v64.grad)):
        assert torch.isfinite(got).all()
        assert rel_l2(got.double().cpu(), ref) < tol and rel_l2(got2.double().cpu(), ref) < tol
        assert rel_l2(got.float(), got2.float()) < tol


def test_attention_softmax_spike(dvt, device):
    """Online-softmax rescale path: one key dominates late in the sequence."""
    g = torch.Generator().manual_seed(10)
    B, H, L, dh = 1, 1, 96, 64
    q = torch.randn(B, H, L, dh, generator=g)
    k = torch.randn(B, H, L, dh, generator=g)
    v = torch.randn(B, H, L, dh, generator=g)
    k[0, 0, 70] = 6.0 * q[0, 0, 3]       # spikes row 3 at key 70 (third 32-key step)
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    ref = O.attention_core(qb.float(), kb.float(), vb.float(), dh ** -0.5)
    out = dvt.functional.attention_core(qb.cuda(), kb.cuda(), vb.cuda(), dh ** -0.5)
    assert rel_l2(out, ref) < BF16_TOL
    assert torch.isfinite(out).all()


# ------------------------------------------------------------------ fused residual blocks
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dim,heads,dh,N", [(64, 2, 32, 10), (128, 2, 64, 17), (64, 1, 64, 9)])
def test_attn_and_mlp_blocks(dvt, device, dtype, dim, heads, dh, N):
    g = torch.Generator().manual_seed(11)
    S = 3
    inner = heads * dh
    project = not (heads == 1 and dh == dim)
    x_d, x = _rnd((S, N, dim), dtype, g)
    P = {
        "ln_w": 1 + 0.1 * torch.randn(dim, generator=g), "ln_b": 0.1 * torch.randn(dim, generator=g),
        "wqkv": torch.randn(3 * inner, dim, generator=g) / math.sqrt(dim),
        "wout": torch.randn(dim, inner, generator=g) / math.sqrt(inner) if project else None,
        "bout": 0.1 * torch.randn(dim, generator=g) if project else None,
        "w1": torch.randn(4 * dim, dim, generator=g) / math.sqrt(dim), "b1": 0.1 * torch.randn(4 * dim, generator=g),
        "w2": torch.randn(dim, 4 * dim, generator=g) / math.sqrt(4 * dim), "b2": 0.1 * torch.randn(dim, generator=g),
    }
    rd = lambda t: None if t is None else t.to(dtype).float()      # weights as the kernels see them
    R = {k: (None if v is None else (rd(v) if v.dim() == 2 else v).clone().requires_grad_(True)) for k, v in P.items()}
    D = {k: (None if v is None else v.cuda().requires_grad_(True)) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    xd = x_d.clone().requires_grad_(True)
    F = dvt.functional
    ref_a = O.self_attention(O.layernorm(xr, R["ln_w"], R["ln_b"]), R["wqkv"], R["wout"], R["bout"], heads) + xr
    out_a = F.attn_block(xd, D["ln_w"], D["ln_b"], D["wqkv"], D["wout"], D["bout"], heads)
    tol = _tol(dtype)
    assert rel_l2(out_a, ref_a) < tol
    ref_m = O.feedforward(O.layernorm(ref_a, R["ln_w"], R["ln_b"]), R["w1"], R["b1"], R["w2"], R["b2"]) + ref_a
    out_m = F.mlp_block(out_a, D["ln_w"], D["ln_b"], D["w1"], D["b1"], D["w2"], D["b2"])
    assert rel_l2(out_m, ref_m) < 2 * tol
    gy_d, gy = _rnd(out_m.shape, dtype, g)
    ref_m.backward(gy)
    out_m.backward(gy_d)
    assert rel_l2(xd.grad, xr.grad) < 3 * tol
    for k in P:
        if P[k] is not None:
            assert rel_l2(D[k].grad, R[k].grad) < 3 * tol, k


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dim,heads,dh,N,S", [(128, 2, 64, 17, 5), (512, 8, 64, 197, 6), (64, 2, 32, 10, 3),
                                              (128, 2, 64, 33, 8)])
def test_attn_block_cls_equals_dense_block_row0(dvt, device, dtype, dim, heads, dh, N, S):
    """F.attn_block_cls(x) == (attention block of vit.py:71-73)(x)[:, 0] (vit.py:119-120 reads only that row), values
    and every gradient: the oracle runs the dense block and slices."""
    g = torch.Generator().manual_seed(21)
    inner = heads * dh
    x_d, x = _rnd((S, N, dim), dtype, g)
    P = {
        "ln_w": 1 + 0.1 * torch.randn(dim, generator=g), "ln_b": 0.1 * torch.randn(dim, generator=g),
        "wqkv": torch.randn(3 * inner, dim, generator=g) / math.sqrt(dim),
        "wout": torch.randn(dim, inner, generator=g) / math.sqrt(inner), "bout": 0.1 * torch.randn(dim, generator=g),
    }
    rd = lambda t: t.to(dtype).float()
    R = {k: (rd(v) if v.dim() == 2 else v).clone().requires_grad_(True) for k, v in P.items()}
    D = {k: v.cuda().requires_grad_(True) for k, v in P.items()}
    xr, xd = x.clone().requires_grad_(True), x_d.clone().requires_grad_(True)
    ref = (O.self_attention(O.layernorm(xr, R["ln_w"], R["ln_b"]), R["wqkv"], R["wout"], R["bout"], heads) + xr)[:, 0]
    out = dvt.functional.attn_block_cls(xd, D["ln_w"], D["ln_b"], D["wqkv"], D["wout"], D["bout"], heads)
    tol = _tol(dtype)
    assert out.shape == (S, dim) and rel_l2(out, ref) < tol
    gy_d, gy = _rnd(out.shape, dtype, g)
    ref.backward(gy)
    out.backward(gy_d)
    assert rel_l2(xd.grad, xr.grad) < 3 * tol
    for k in P:
        assert rel_l2(D[k].grad, R[k].grad) < 3 * tol, k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("dim,heads,dh,N,S", [(128, 2, 64, 17, 5), (512, 8, 64, 197, 6), (64, 2, 32, 10, 3),
                                              (384, 6, 64, 197, 3), (512, 8, 64, 200, 2), (512, 3, 64, 1, 9),
                                              (512, 8, 64, 325, 2), (128, 2, 64, 201, 3)])
def test_attn_block_cls_folded_equals_dense_block_row0(dvt, device, dtype, dim, heads, dh, N, S, monkeypatch):
    """The same contract with the K / V projections folded into the one query (csrc/attention_cls.hip: no LN(x), K, V
    of the rows that are never read again): values and every gradient against the dense block + slice of the oracle."""
    monkeypatch.setattr(dvt.functional, "CLS_FOLD_MIN_ROWS", 0)
    g = torch.Generator().manual_seed(23)
    inner = heads * dh
    x_d, x = _rnd((S, N, dim), dtype, g)
    assert dvt.ops.attn_cls_supported(x_d, heads)
    P = {
        "ln_w": 1 + 0.1 * torch.randn(dim, generator=g), "ln_b": 0.1 * torch.randn(dim, generator=g),
        "wqkv": torch.randn(3 * inner, dim, generator=g) / math.sqrt(dim),
        "wout": torch.randn(dim, inner, generator=g) / math.sqrt(inner), "bout": 0.1 * torch.randn(dim, generator=g),
    }
    rd = lambda t: t.to(dtype).float()
    R = {k: (rd(v) if v.dim() == 2 else v).clone().requires_grad_(True) for k, v in P.items()}
    D = {k: v.cuda().requires_grad_(True) for k, v in P.items()}
    xr, xd = x.clone().requires_grad_(True), x_d.clone().requires_grad_(True)
    ref = (O.self_attention(O.layernorm(xr, R["ln_w"], R["ln_b"]), R["wqkv"], R["wout"], R["bout"], heads) + xr)[:, 0]
    out = dvt.functional.attn_block_cls(xd, D["ln_w"], D["ln_b"], D["wqkv"], D["wout"], D["bout"], heads)
    assert out.shape == (S, dim) and rel_l2(out, ref) < BF16_TOL
    gy_d, gy = _rnd(out.shape, dtype, g)
    ref.backward(gy)
    out.backward(gy_d)
    assert rel_l2(xd.grad, xr.grad) < 3 * BF16_TOL
    for k in P:
        assert rel_l2(D[k].grad, R[k].grad) < 3 * BF16_TOL, k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("S,N,d,H", [(4, 197, 512, 8), (3, 50, 384, 6), (2, 9, 64, 2), (5, 200, 512, 1), (3, 325, 512, 8),
                                     (2, 400, 512, 8), (2, 201, 128, 3)])
def test_attn_cls_folded_kernels_against_float64(dvt, device, dtype, S, N, d, H):
    """dvt_attn_cls_fwd / _bwd on 16-bit rows: everything between the rows and the fp32 results is fp32, so the fp32
    outputs are held to 2e-5 against a float64 evaluation of the same formulas on the same inputs; dx (16-bit) to the
    rounding of its element type.  Also: dgamma / dbeta accumulate flags, strided rows."""
    g = torch.Generator().manual_seed(24)
    xs = torch.randn(S, N + 2, d, generator=g).to(dtype)               # rows sit in a larger buffer: sequence stride (N+2) d
    x_d = xs.cuda()[:, 1:N + 1]
    x = xs[:, 1:N + 1].double()
    gam, bet = 1 + 0.2 * torch.randn(d, generator=g), 0.2 * torch.randn(d, generator=g)
    R = torch.randn(S, H, d, generator=g) * (1.5 / math.sqrt(d))
    dM = torch.randn(S, H, d, generator=g)
    eps = 1e-5
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rs = (var + eps).rsqrt()
    n = ((x - mu) * rs).requires_grad_(True)
    gd, bd, Rd = (t.double().requires_grad_(True) for t in (gam, bet, R))
    xh = n * gd + bd
    s = torch.einsum("shd,snd->snh", Rd, xh)
    p = torch.softmax(s, dim=1)
    M = torch.einsum("snh,snd->shd", p, xh)
    A_ref = torch.einsum("snh,snd->shd", p, n)
    A, lse, P, mean, rstd = dvt.ops.attn_cls_fwd(x_d, gam.cuda(), bet.cuda(), eps, R.cuda())
    assert rel_l2(P[:, :, :H], p.detach()) < 2e-5
    assert rel_l2(A, A_ref.detach()) < 2e-5
    assert rel_l2(lse, torch.logsumexp(s, dim=1).detach()) < 2e-6
    assert rel_l2(mean.view(S, N), mu[..., 0]) < 2e-5 and rel_l2(rstd.view(S, N), rs[..., 0]) < 2e-5
    M.backward(dM.double())
    # the LayerNorm backward of d LN(x) = n.grad (already multiplied by gamma on the way down)
    dn = n.grad
    dx_ref = rs * (dn - dn.mean(-1, keepdim=True) - n.detach() * (dn * n.detach()).mean(-1, keepdim=True))
    G_ref = Rd.grad / gd.detach()                                      # dr_h = gamma G_h
    dg0 = torch.full((d,), 2.0, device="cuda")
    db0 = torch.full((d,), 3.0, device="cuda")
    dx, G, dg, db = dvt.ops.attn_cls_bwd(x_d, gam.cuda(), bet.cuda(), eps, R.cuda(), A, lse, P, mean, rstd, dM.cuda(),
                                         dg=dg0, db=db0, accumulate=True, accumulate_beta=False)
    assert dx.stride() == x_d.stride()
    unit = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert rel_l2(dx, dx_ref) < unit
    assert rel_l2(G, G_ref) < 5e-5
    assert rel_l2(dg - 2.0, gd.grad) < 5e-5 and rel_l2(db, bd.grad) < 5e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("S,H,dh,d", [(13, 8, 64, 512), (256, 8, 64, 512), (5, 2, 32, 64), (9, 6, 64, 384), (3, 1, 96, 128)])
def test_heads_products(dvt, device, dtype, S, H, dh, d):
    """dvt_heads_expand / _contract / _outer (per-head products of the folded single-query attention) against einsum in
    float64; the weight is a row range of a larger packed matrix (row stride d, offset rows)."""
    g = torch.Generator().manual_seed(25)
    inner = H * dh
    Wp_d, Wp = _rnd((3 * inner, d), dtype, g, 1 / math.sqrt(d))
    W_d, W = Wp_d[inner:2 * inner], Wp[inner:2 * inner].double().view(H, dh, d)
    a_d, a = _rnd((S, inner + 8), dtype, g)
    a_d, a = a_d[:, :inner], a[:, :inner].double().view(S, H, dh)
    v = torch.randn(S, H, d, generator=g)
    gam, bet = 1 + 0.2 * torch.randn(d, generator=g), 0.2 * torch.randn(d, generator=g)
    unit = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    out = dvt.ops.heads_expand(a_d, W_d, H, 0.37)
    assert rel_l2(out, 0.37 * torch.einsum("she,hed->shd", a, W)) < 2e-6
    vv = v.double() * gam.double() + bet.double()
    out = dvt.ops.heads_contract(v.cuda(), W_d, 1.7, gam.cuda(), bet.cuda())
    assert out.dtype == dtype and rel_l2(out, 1.7 * torch.einsum("shd,hed->she", vv, W).reshape(S, inner)) < unit
    out = dvt.ops.heads_contract(v.cuda(), W_d, 1.0)
    assert rel_l2(out, torch.einsum("shd,hed->she", v.double(), W).reshape(S, inner)) < unit
    buf = torch.full((3 * inner, d), 0.5, device="cuda")
    dvt.ops.heads_outer(a_d, v.cuda(), buf[inner:2 * inner], 0.9, gam.cuda(), bet.cuda(), accumulate=True)
    ref = 0.9 * torch.einsum("she,shd->hed", a, vv).reshape(inner, d)
    assert rel_l2(buf[inner:2 * inner] - 0.5, ref) < 5e-6
    assert (buf[:inner] == 0.5).all() and (buf[2 * inner:] == 0.5).all()
    dvt.ops.heads_outer(a_d, v.cuda(), buf[inner:2 * inner], 1.0, gam.cuda())
    assert rel_l2(buf[inner:2 * inner], torch.einsum("she,shd->hed", a, v.double() * gam.double()).reshape(inner, d)) < 5e-6
    # the pairs the backward launches together: bit-identical to their members
    b1, b2 = torch.full((3 * inner, d), 0.5, device="cuda"), torch.full((3 * inner, d), 0.5, device="cuda")
    e1 = dvt.ops.heads_expand_outer(a_d, W_d, H, v.cuda(), b1[inner:2 * inner], alpha_out=0.37, alpha_dw=0.9, gamma=gam.cuda(),
                                    beta=bet.cuda(), accumulate=True)
    dvt.ops.heads_outer(a_d, v.cuda(), b2[inner:2 * inner], 0.9, gam.cuda(), bet.cuda(), accumulate=True)
    assert torch.equal(e1, dvt.ops.heads_expand(a_d, W_d, H, 0.37)) and torch.equal(b1, b2)
    c1 = dvt.ops.heads_contract_outer(v.cuda(), gam.cuda(), W_d, a_d, b1[:inner], alpha_out=1.7, alpha_dw=0.3)
    dvt.ops.heads_outer(a_d, v.cuda(), b2[:inner], 0.3, gam.cuda())
    assert torch.equal(c1, dvt.ops.heads_contract(v.cuda(), W_d, 1.7, gam.cuda())) and torch.equal(b1, b2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_bwd_first_row_operands(dvt, device, dtype):
    """dvt_layernorm_bwd_first: dy_first enters dy of row (i0, 0), dx_first enters dx of that row; dgamma / dbeta
    accumulate independently."""
    g = torch.Generator().manual_seed(22)
    S, N, d = 5, 7, 128
    x_d, x = _rnd((S, N, d), dtype, g)
    dy_d, dy = _rnd((S, N, d), dtype, g)
    f_d, f = _rnd((S, d), dtype, g)
    r_d, r = _rnd((S, d), dtype, g)
    w = 1 + 0.1 * torch.randn(d, generator=g)
    b = 0.1 * torch.randn(d, generator=g)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    dy_tot = dy.clone()
    dy_tot[:, 0] += f
    O.layernorm(xr, wr, br).backward(dy_tot)
    dx_ref = xr.grad.clone()
    dx_ref[:, 0] += r
    _, mean, rstd = dvt.ops.layernorm_fwd(x_d.view(-1, d), w.cuda(), b.cuda())
    dg0 = torch.full((d,), 2.0, device="cuda")
    db0 = torch.full((d,), 3.0, device="cuda")
    dx, dg, db = dvt.ops.layernorm_bwd(dy_d.view(-1, d), x_d.view(-1, d), w.cuda(), mean, rstd, rows=(S, N, N * d, d),
                                       dy_first=f_d, dx_first=r_d, dg=dg0, db=db0, accumulate=True,
                                       accumulate_beta=False)
    tol = _tol(dtype)
    assert rel_l2(dx.view(S, N, d), dx_ref) < 2 * tol
    assert rel_l2(dg - 2.0, wr.grad) < 2 * tol and rel_l2(db, br.grad) < 2 * tol


# ------------------------------------------------------------------ losses / optimizer
@pytest.mark.parametrize("dtype", DTYPES)
def test_losses(dvt, device, dtype):
    g = torch.Generator().manual_seed(12)
    z_d, z = _rnd((8, 19), dtype, g, 2.0)
    y = (torch.rand(8, 19, generator=g) < 0.2).float()
    zr = z.clone().requires_grad_(True)
    ref = O.bce_with_logits(zr, y)
    zd = z_d.clone().requires_grad_(True)
    out = dvt.functional.bce_with_logits(zd, y.cuda())
    assert abs(float(out) - float(ref)) < 1e-5
    ref.backward()
    out.backward()
    assert rel_l2(zd.grad, zr.grad) < _tol(dtype)
    t_d, t = _rnd((8, 19), dtype, g)
    zr2 = z.clone().requires_grad_(True)
    ref2 = O.cross_entropy_hard(zr2, t)
    zd2 = z_d.clone().requires_grad_(True)
    out2 = dvt.functional.cross_entropy_argmax(zd2, t_d)
    assert abs(float(out2) - float(ref2)) < 1e-5
    ref2.backward()
    out2.backward()
    assert rel_l2(zd2.grad, zr2.grad) < _tol(dtype)


def test_adamw_matches_torch(dvt, device):
    g = torch.Generator().manual_seed(13)
    p = torch.randn(1000, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=5e-3, weight_decay=0.09)      # frame_transformer.py:127-129
    pd, m, v = p.cuda(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        pr.grad = gr.clone()
        opt.step()
        dvt.ops.adamw_step_(pd, gr.cuda(), m, v, lr=5e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.09,
                            step=step)
    assert rel_l2(pd, pr) < 1e-6
    # device-side step counter variant (hipGraph-capturable)
    p2, m2, v2 = p.cuda(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    sd = torch.zeros(1, dtype=torch.int64, device="cuda")
    g2 = torch.Generator().manual_seed(13)
    torch.randn(1000, generator=g2)
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g2)
        dvt.ops.adamw_step_dev_(p2, gr.cuda(), m2, v2, sd, lr=5e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.09)
    assert int(sd) == 3 and rel_l2(p2, pr) < 1e-6
    # the fused step of the training loop: update + 16-bit mirror + counter in one launch (also a length that is not a
    # multiple of 4, a skip mask, and enough elements for many workgroups to take the ticket)
    for n, dt in ((1000, torch.bfloat16), (1003, torch.float16), (300001, None)):
        g3 = torch.Generator().manual_seed(14)
        p0 = torch.randn(n, generator=g3)
        pr3 = p0.clone().requires_grad_(True)
        opt3 = torch.optim.AdamW([pr3], lr=5e-3, weight_decay=0.09)
        p3, m3, v3 = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        sd2 = torch.zeros(2, dtype=torch.int64, device="cuda")
        mirror = torch.empty(n, dtype=dt, device="cuda") if dt is not None else None
        skip = torch.zeros((n + 63) // 64, dtype=torch.uint8)
        skip[1] = 1                                      # elements 64..127: no gradient this run -> untouched
        for step in range(3):
            gr = torch.randn(n, generator=g3)
            pr3.grad = gr.clone()
            keep = pr3.detach()[64:128].clone()
            opt3.step()
            with torch.no_grad():
                pr3[64:128] = keep
            dvt.ops.adamw_step_fused_(p3, gr.cuda(), m3, v3, sd2, lr=5e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                                      weight_decay=0.09, skip=skip.cuda(), mirror=mirror)
        assert sd2.tolist() == [3, 0] and rel_l2(p3, pr3) < 1e-6
        assert torch.equal(p3[64:128].cpu(), p0[64:128]) and float(m3[64:128].abs().max()) == 0.0
        if mirror is not None:
            assert torch.equal(mirror, p3.to(dt))


@pytest.mark.parametrize("kind", ["sgd", "sgd_nomom", "adagrad", "adamW"])
def test_optimizer_classes_match_torch(dvt, device, kind):
    """configure_optimizers() choices of the reference (frame_transformer.py:123-134) behind the
    torch.optim.Optimizer interface, against torch's own CPU optimizers (3 steps, 2 tensors)."""
    from dvt_amd import optim
    g = torch.Generator().manual_seed(17)
    shapes = [(37, 19), (130,)]
    ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    got = [r.detach().clone().cuda().requires_grad_(True) for r in ref]
    if kind == "sgd":
        o_ref = torch.optim.SGD(ref, lr=5e-2, momentum=0.005, weight_decay=0.09)          # config.yaml:10-12
        o_got = optim.SGD(got, lr=5e-2, momentum=0.005, weight_decay=0.09)
    elif kind == "sgd_nomom":
        o_ref = torch.optim.SGD(ref, lr=5e-2, weight_decay=0.01)
        o_got = optim.SGD(got, lr=5e-2, weight_decay=0.01)
    elif kind == "adagrad":
        o_ref = torch.optim.Adagrad(ref, lr=5e-2, weight_decay=0.09)
        o_got = optim.Adagrad(got, lr=5e-2, weight_decay=0.09)
    else:
        o_ref = torch.optim.AdamW(ref, lr=5e-3, weight_decay=0.09)
        o_got = optim.AdamW(got, lr=5e-3, weight_decay=0.09)
    for _ in range(3):
        for r, t in zip(ref, got):
            gr = torch.randn(r.shape, generator=g)
            r.grad, t.grad = gr.clone(), gr.cuda()
        o_ref.step()
        o_got.step()
    for r, t in zip(ref, got):
        assert rel_l2(t, r) < 1e-6
    sd = o_got.state_dict()                                    # torch-compatible state layout
    assert set(sd) == {"state", "param_groups"}


def test_fp16_loss_scaling_skips_overflow_and_recovers(dvt, device):
    """BASELINE configs[4] (fp16 + loss scaling): the scaler lives on the device.  A step whose gradient overflows
    is skipped and halves the scale; clean steps apply AdamW on grad / scale and grow the scale every interval."""
    from dvt_amd.dp import FlatParameters
    lin = torch.nn.Linear(16, 8).cuda()
    flat = FlatParameters(lin, compute_dtype=torch.float16)
    seed = flat.enable_loss_scaling(init_scale=1024.0, growth_interval=2)
    assert float(seed) == 1024.0
    ref = torch.nn.Linear(16, 8)
    ref.load_state_dict({k: v.detach().cpu() for k, v in lin.state_dict().items()})
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.09)
    g = torch.Generator().manual_seed(0)

    def step(grads, overflow=False):
        scale = float(flat.scale_dev)
        flat.zero_grad()
        for p, gr in zip(lin.parameters(), grads):
            buf = p._dvt_sink.buf
            buf.copy_((gr * scale).cuda())
            if overflow:
                buf.view(-1)[0] = float("inf")
            p._dvt_sink.mark_written()
        flat.finish_backward()
        flat.adamw_step(lr=1e-2, weight_decay=0.09)

    grads = [torch.randn(p.shape, generator=g) for p in ref.parameters()]
    before = flat.data.clone()
    step(grads, overflow=True)
    assert torch.equal(flat.data, before) and float(flat.scale_dev) == 512.0 and int(flat.step_dev[0]) == 0
    assert float(flat.loss_grad) == 512.0 and int(flat.found_inf) == 0
    for _ in range(2):
        grads = [torch.randn(p.shape, generator=g) for p in ref.parameters()]
        for p, gr in zip(ref.parameters(), grads):
            p.grad = gr.clone()
        opt.step()
        step(grads)
    assert int(flat.step_dev[0]) == 2 and float(flat.scale_dev) == 1024.0          # grew after 2 clean steps
    for p, q in zip(lin.parameters(), ref.parameters()):
        assert rel_l2(p, q) < 1e-6
    w16 = lin.weight._dvt_compute
    assert w16.dtype == torch.float16 and torch.equal(w16.float().cpu(), lin.weight.detach().cpu().half().float())


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_philox(dvt, device, dtype):
    """nn.Dropout in training mode: Philox mask from (device state, site offset, index); survivors scaled by
    1 / (1 - p); backward re-draws the same mask; a new step (device-side advance) draws a new one."""
    F = dvt.functional
    F.manual_seed(77)
    x = torch.randn(1 << 16, generator=torch.Generator().manual_seed(1)).to(dtype).cuda().requires_grad_(True)
    y = F.dropout(x, 0.3, True)
    kept = y != 0
    assert abs(float(kept.float().mean()) - 0.7) < 0.01
    assert torch.allclose(y[kept].float(), (x.detach()[kept].float() / 0.7), rtol=1e-2 if dtype != torch.float32 else 1e-6)
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad != 0, kept)                                  # same mask in backward
    y2 = F.dropout(x.detach(), 0.3, True)                                  # next site of the same step: new mask
    assert not torch.equal(y2 != 0, kept)
    F.manual_seed(77)
    y3 = F.dropout(x.detach(), 0.3, True)                                  # same seed, same site: same mask
    assert torch.equal(y3, y.detach())
    F.next_step()                                                          # device-side advance
    y4 = F.dropout(x.detach(), 0.3, True)
    assert not torch.equal(y4 != 0, kept)
    assert F.dropout(x, 0.3, False) is x and F.dropout(x, 0.0, True) is x   # eval / p = 0: identity
    # the 4 Philox words of a block are independent: no structure at stride 4
    k4 = kept.view(-1, 4).float()
    assert float((k4[:, 0] * k4[:, 1]).mean()) == pytest.approx(0.49, abs=0.02)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_fused_with_residual_and_relu(dvt, device, dtype):
    """dvt_dropout_fused: res + dropout(x) and dropout(relu(x)) (the training-mode sites of nn.TransformerEncoderLayer,
    frame_transformer.py:39-47) in one launch each draw the SAME mask as dvt_dropout at the same RNG site, and their backward
    passes equal the unfused compositions'."""
    F = dvt.functional
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(28, 896, generator=g).to(dtype).cuda()
    r0 = torch.randn(28, 896, generator=g).to(dtype).cuda()
    gy = torch.randn(28, 896, generator=g).to(dtype).cuda()
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    out = {}
    for fused in (False, True):
        F.manual_seed(11)
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        a = F.dropout_add(x, r, 0.5, True) if fused else F.add(r, F.dropout(x, 0.5, True))
        x2 = x0.clone().requires_grad_(True)
        b = F.relu_dropout(x2, 0.5, True) if fused else F.dropout(F.relu(x2), 0.5, True)
        torch.autograd.backward([a, b], [gy, gy])
        out[fused] = (a.detach().float(), b.detach().float(), x.grad.float(), r.grad.float(), x2.grad.float())
    for u, f in zip(out[False], out[True]):
        assert float((u - f).abs().max()) <= tol * max(1.0, float(u.abs().max()))
    a, b = out[True][0], out[True][1]
    assert abs(float(((a - r0.float()) != 0).float().mean()) - 0.5) < 0.03          # half of the elements survive
    assert torch.equal(out[True][1] != 0, out[False][1] != 0)                         # the same mask
    assert torch.equal(F.dropout_add(x0, r0, 0.5, False), F.add(r0, x0)) and torch.equal(F.relu_dropout(x0, 0.0, True), F.relu(x0))


def test_dropout_training_step_and_checkpoint_consistency(dvt, device):
    """ViViT with dropout > 0 in train mode runs end to end, and activation checkpointing replays the same masks
    (gradients identical with and without recomputation)."""
    from dvt_amd.models.vit import ViViT
    F = dvt.functional
    res = {}
    for ck in (False, True):
        torch.manual_seed(3)
        F.manual_seed(5)
        net = ViViT(32, 8, 19, 3, dim=64, depth=2, heads=2, dim_head=32, dropout=0.2, emb_dropout=0.1,
                    compute_dtype=torch.float32, activation_checkpointing=ck).cuda().train()
        x = torch.randn(2, 3, 3, 32, 32, generator=torch.Generator().manual_seed(4)).cuda()
        out = net(x)
        out.square().mean().backward()
        res[ck] = (out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()})
    assert torch.equal(res[False][0], res[True][0])
    for k in res[False][1]:
        assert torch.equal(res[False][1][k], res[True][1][k]), k
    ev = net.eval()(x)
    assert not torch.equal(ev, res[True][0])                               # dropout really was active


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_probability_dropout(dvt, device, dtype):
    """nn.MultiheadAttention(dropout=p) in training mode: o = (softmax(s) * keep / (1-p)) v with a Philox mask that
    forward and both backward kernels re-draw identically.  The mask is recovered from the kernel itself (v = identity
    gives the dropped probabilities) and the outputs / gradients are then checked against autograd on that mask."""
    ops, F = dvt.ops, dvt.functional
    B, H, L, dh = 2, 2, 24, 32
    g = torch.Generator().manual_seed(9)
    F.manual_seed(5)
    st = F._rng.tensor(torch.device("cuda"))
    q, k = (torch.randn(B, H, L, dh, generator=g).to(dtype).cuda() for _ in range(2))
    p = 0.4
    # probabilities with dropout: use v = [I | 0] (dh >= L would be needed in general; L = 24 <= 32 here)
    v_eye = torch.zeros(B, H, L, dh, dtype=dtype, device="cuda")
    v_eye[:, :, torch.arange(L), torch.arange(L)] = 1
    o = torch.empty(B, H, L, dh, dtype=dtype, device="cuda")
    ops.attention_fwd(q, k, v_eye, o, dh ** -0.5, (p, st, 0))
    pd = o[..., :L].float().cpu()                                        # dropped + rescaled probabilities
    sm = torch.softmax((q.float() @ k.float().transpose(-1, -2)).cpu() * dh ** -0.5, -1)
    mask = (pd > 0).float()
    assert abs(float(mask.mean()) - (1 - p)) < 0.06
    assert torch.allclose(pd, sm * mask / (1 - p), atol=2e-2 if dtype != torch.float32 else 1e-5)
    # full forward / backward against autograd with that mask
    v = torch.randn(B, H, L, dh, generator=g).to(dtype).cuda()
    lse = ops.attention_fwd(q, k, v, o, dh ** -0.5, (p, st, 0))
    qr, kr, vr = (t.float().cpu().requires_grad_(True) for t in (q, k, v))
    ref = (torch.softmax(qr @ kr.transpose(-1, -2) * dh ** -0.5, -1) * mask / (1 - p)) @ vr
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel_l2(o, ref) < tol
    do = torch.randn(B, H, L, dh, generator=g).to(dtype)
    ref.backward(do.float())
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    ops.attention_bwd(q, k, v, o, lse, do.cuda(), dq, dk, dv, dh ** -0.5, (p, st, 0))
    assert rel_l2(dq, qr.grad) < 3 * tol and rel_l2(dk, kr.grad) < 3 * tol and rel_l2(dv, vr.grad) < 3 * tol


def test_abi_error_reporting_and_edge_sizes(dvt, device):
    """Every entry point returns a status and leaves a message in dvt_last_error(); nothing is launched on bad
    arguments.  Also the degenerate sizes the reference can produce (empty batch, one row, one token)."""
    import ctypes as C
    L = dvt._lib
    lib = L.load()
    ops = dvt.ops
    x = torch.randn(8, 16, device="cuda")
    # null pointers / bad sizes straight through the C ABI
    assert lib.dvt_cast(None, L.F32, x.data_ptr(), L.BF16, 8, None) != 0
    assert b"dvt_cast" in lib.dvt_last_error()
    assert lib.dvt_dropout(x.data_ptr(), x.data_ptr(), 8, C.c_float(1.5), x.data_ptr(), 0, L.F32, None) != 0
    assert lib.dvt_layernorm_fwd(x.data_ptr(), None, None, x.data_ptr(), None, None, 8, 1, 16, 16, 0, 16, 0, C.c_float(1e-5),
                                 L.F32, None) != 0
    assert lib.dvt_adamw_step(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 8, C.c_float(1e-3), C.c_float(0.9),
                              C.c_float(0.999), C.c_float(1e-8), C.c_float(0.0), 0, None) != 0      # step counts from 1
    with pytest.raises(RuntimeError, match="dvt_"):
        L.check(lib.dvt_frames_preprocess(x.data_ptr(), x.data_ptr(), L.F32, 1, 4, 4, 0, 2, x.data_ptr(), x.data_ptr(),
                                          x.data_ptr(), None), "dvt_frames_preprocess")
    # unsupported dtype code
    assert lib.dvt_add(x.data_ptr(), x.data_ptr(), x.data_ptr(), 8, 7, None) != 0
    # degenerate sizes
    assert ops.cast(torch.empty(0, device="cuda"), torch.bfloat16).numel() == 0
    one = torch.randn(1, 16, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(one, torch.ones(16, device="cuda"), torch.zeros(16, device="cuda"))
    assert torch.allclose(y.cpu(), torch.nn.functional.layer_norm(one.cpu(), (16,)), atol=1e-5)
    w = torch.randn(3, 16, device="cuda")
    assert rel_l2(ops.linear_fwd(one, w), one.cpu() @ w.cpu().t()) < 1e-5
    q = torch.randn(1, 1, 1, 8, device="cuda")                       # one token attends to itself: o == v
    o = torch.empty_like(q)
    ops.attention_fwd(q, q, q, o, 1.0)
    assert torch.allclose(o, q, atol=1e-6)
    # wrong device is refused before anything is launched
    with pytest.raises(RuntimeError, match="GPU"):
        ops.layernorm_fwd(one.cpu(), torch.ones(16), torch.zeros(16))


@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d,C,ln1", [(8, 512, 19, True), (1, 64, 1, True), (32, 1024, 12, False), (5, 192, 7, True), (9, 768, 16, True)])
def test_head_and_loss_in_one_launch(device, xdt, rows, d, C, ln1):
    """dvt_head_bce_fwd + dvt_scaled_emit_group (vit.py:97-100,126-128 + BCEWithLogitsLoss): loss, logits and every
    gradient against torch autograd in fp32 on the same (rounded) input, with a non-trivial upstream gradient, an
    accumulating destination and the 16-bit copy of dx."""
    from dvt_amd import ops
    g = torch.Generator().manual_seed(rows * 1000 + d + C)
    x = torch.randn(rows, d, generator=g).to(xdt)
    P = [1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g), 1 + 0.1 * torch.randn(d, generator=g),
         0.1 * torch.randn(d, generator=g), torch.randn(C, d, generator=g) / d ** 0.5, 0.1 * torch.randn(C, generator=g)]
    t = (torch.rand(rows, C, generator=g) < 0.3).float()
    xr = x.float().clone().requires_grad_(True)
    Pr = [p.clone().requires_grad_(True) for p in P]
    h = torch.nn.functional.layer_norm(xr, (d,), Pr[0], Pr[1], 1e-5) if ln1 else xr
    z = torch.nn.functional.linear(torch.nn.functional.layer_norm(h, (d,), Pr[2], Pr[3], 1e-6), Pr[4], Pr[5])
    ref = torch.nn.functional.binary_cross_entropy_with_logits(z, t)
    ref.backward(torch.tensor(3.0))
    Pc = [p.cuda() for p in P]
    loss, logits, G = ops.head_bce_fwd(x.cuda(), Pc[0] if ln1 else None, Pc[1] if ln1 else None, 1e-5, Pc[2], Pc[3], 1e-6,
                                       Pc[4], Pc[5], t.cuda())
    assert abs(float(loss) - float(ref)) < 1e-5 and rel_l2(logits, z.detach()) < 1e-5
    scale = torch.tensor([3.0], device="cuda")
    dx = torch.empty(rows, d, device="cuda")
    dx_lp = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
    old = torch.randn(C, d, generator=g).cuda()
    dw = old.clone()
    names = (["g1", "b1"] if ln1 else []) + ["g2", "b2", "c"]
    outs = {n: torch.empty_like(G[n]) for n in names}
    ops.scaled_emit_group(scale, [(G["x"], dx, False, dx_lp), (G["w"].reshape(-1), dw.view(-1), True, None)]
                          + [(G[n], outs[n], False, None) for n in names])
    assert rel_l2(dx, xr.grad) < 2e-5 and torch.equal(dx_lp, dx.to(torch.bfloat16))
    assert rel_l2(dw - old, Pr[4].grad) < 2e-5
    want = dict(zip(["g1", "b1", "g2", "b2", "w", "c"], [p.grad for p in Pr]))
    for n in names:
        assert rel_l2(outs[n], want[n]) < 2e-5, n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols,stride", [(2, 451584, 451584 * 14), (16, 8192, 8192), (5, 40, 48), (300, 512, 512), (1, 16384, 16384)])
def test_rows_sum_few_and_many_rows(device, dtype, rows, cols, stride):
    """dvt_rows_sum: out[c] (+)= sum_r src[r * stride + c] -- the block-per-chunk form and the thread-per-chunk form for few
    rows of many columns (the pixel-space CLS chunk's gradient over the batch, frame_transformer.py:105,195)."""
    from dvt_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    buf = torch.randn((rows - 1) * stride + cols, generator=g).to(dtype).cuda()
    view = torch.as_strided(buf, (rows, cols), (stride, 1)).float()
    out = ops.rows_sum(buf, stride, rows, cols)
    assert rel_l2(out, view.sum(0)) < 2e-6
    prev = torch.randn(cols, generator=g).cuda()
    acc = prev.clone()
    ops.rows_sum(buf, stride, rows, cols, out=acc, accumulate=True)
    assert rel_l2(acc, prev + view.sum(0)) < 2e-6
