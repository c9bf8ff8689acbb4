"""Parity of the per-frame CNN encoder on the GPU: im2col + MFMA GEMM +
BatchNorm/ReLU/max-pool kernels vs the golden vectors from the imported reference
(tests/golden/resnet18_pyramid.npz) and vs the CPU oracle on small shapes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from tests.util import assert_within_reference_lowprec, golden, rel_l2, fill_resnet_from_numpy

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def dvt():
    import dvt_amd
    return dvt_amd


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("k,stride,pad,Cin,Cout,H", [(3, 1, 1, 16, 24, 10), (3, 2, 1, 8, 16, 9), (1, 2, 0, 16, 32, 8),
                                                    (1, 1, 0, 16, 8, 6), (7, 2, 3, 3, 16, 20)])
def test_conv_bn_relu_block(dvt, device, dtype, tol, k, stride, pad, Cin, Cout, H):
    g = torch.Generator().manual_seed(31)
    N, W = 3, H + 2
    nchw_in = Cin == 3                                  # the stem reads raw NCHW frames
    x = torch.randn(N, Cin, H, W, generator=g)
    conv = torch.nn.Conv2d(Cin, Cout, k, stride, pad, bias=False)
    bn = torch.nn.BatchNorm2d(Cout)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
        bn.weight.copy_(1 + 0.1 * torch.randn(Cout, generator=g)); bn.bias.copy_(0.1 * torch.randn(Cout, generator=g))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, Cout, Ho, Wo, generator=g)
    # ---- oracle (CPU, fp32, same rounded inputs)
    xr = x.to(dtype).float().clone().requires_grad_(not nchw_in)
    wr = conv.weight.detach().to(dtype).float().clone().requires_grad_(True)
    gr, br = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    rr = res.to(dtype).float().clone().requires_grad_(True)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    z = TF.conv2d(xr, wr, None, stride, pad)
    ref = torch.relu(TF.batch_norm(z, rm, rv, gr, br, True, 0.1, 1e-5) + rr)
    # ---- device
    convd, bnd = conv.cuda(), bn.cuda().train()
    if nchw_in:
        xd = x.to(dtype).cuda()
    else:
        xd = x.to(dtype).permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda().detach().requires_grad_(True)
    rd = res.to(dtype).permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda().detach().requires_grad_(True)
    y = dvt.functional.conv_bn_act(xd, convd, bnd, (N, Cin, H, W, nchw_in), relu=True, residual=rd, dtype=dtype)
    ref_nhwc = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    assert rel_l2(y, ref_nhwc) < tol
    assert torch.allclose(bnd.running_mean.cpu(), rm, atol=2e-2 if dtype == torch.bfloat16 else 1e-5)
    assert torch.allclose(bnd.running_var.cpu(), rv, atol=2e-2 if dtype == torch.bfloat16 else 1e-5)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy.to(dtype).float())
    y.backward(gy.to(dtype).permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda())
    # (gradients: a ReLU decision that flips against the fp32 oracle -- y within the 16-bit rounding of z of zero -- moves one
    # of ~5,000 live elements of the layer's gradient, 1.4 % of its L2 norm.  Round 5: every route -- implicit epilogue, halo /
    # streamed-weight kernels, split-K reduce, the explicit statistics pass -- sums the STORED 16-bit z, so the bound is the
    # round-3 one again: 2 x tol)
    gtol = 2 * tol
    assert rel_l2(convd.weight.grad, wr.grad) < gtol
    assert rel_l2(bnd.weight.grad, gr.grad) < gtol and rel_l2(bnd.bias.grad, br.grad) < gtol
    assert rel_l2(rd.grad, rr.grad.permute(0, 2, 3, 1).reshape(-1, Cout)) < gtol
    if not nchw_in:
        assert rel_l2(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin)) < gtol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Cout,H,W", [(64, 32, 32), (16, 64, 48)])
def test_stem_as_implicit_gemm_matches_conv2d(dvt, device, dtype, Cout, H, W):
    """The 7x7 / 2 stem on raw NCHW frames (custom_resnet.py:100) through the implicit-GEMM form -- frames zero-extended to
    8 channels (dvt_nchw_to_nhwc_pad), K rounded up to the k-tile, no column matrix -- against torch's conv2d + batch_norm
    on the same rounded operands: output, running statistics, weight / BatchNorm gradients; and against the explicit
    im2col path of the same library."""
    from dvt_amd import functional as F
    g = torch.Generator().manual_seed(41)
    N = 2
    x = torch.randn(N, 3, H, W, generator=g)
    conv = torch.nn.Conv2d(3, Cout, 7, 2, 3, bias=False)
    bn = torch.nn.BatchNorm2d(Cout)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / 147) ** 0.5)
        bn.weight.copy_(1 + 0.1 * torch.randn(Cout, generator=g)); bn.bias.copy_(0.1 * torch.randn(Cout, generator=g))
    wr = conv.weight.detach().to(dtype).float().clone().requires_grad_(True)
    gr, br = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    z = TF.conv2d(x.to(dtype).float(), wr, None, 2, 3)
    ref = torch.relu(TF.batch_norm(z, rm, rv, gr, br, True, 0.1, 1e-5))
    gy = torch.randn(ref.shape, generator=g).to(dtype)
    ref.backward(gy.float())
    outs = {}
    for implicit in (True, False):
        F.IMPLICIT_CONV = implicit
        try:
            c2, b2 = torch.nn.Conv2d(3, Cout, 7, 2, 3, bias=False), torch.nn.BatchNorm2d(Cout)
            c2.load_state_dict(conv.state_dict()); b2.load_state_dict(bn.state_dict())
            c2, b2 = c2.cuda(), b2.cuda().train()
            y = F.conv_bn_act(x.to(dtype).cuda(), c2, b2, (N, 3, H, W, True), relu=True, dtype=dtype)
            y.backward(gy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda())
            outs[implicit] = (y.detach(), c2.weight.grad.clone(), b2.weight.grad.clone(), b2.bias.grad.clone(), b2.running_var.clone())
        finally:
            F.IMPLICIT_CONV = True
    tol = 1.5e-2 if dtype == torch.bfloat16 else 3e-3
    y, dw, dg, db, rvar = outs[True]
    assert rel_l2(y, ref.permute(0, 2, 3, 1).reshape(-1, Cout)) < tol
    assert rel_l2(dw, wr.grad) < 2 * tol and rel_l2(dg, gr.grad) < 2 * tol and rel_l2(db, br.grad) < 2 * tol
    assert torch.allclose(rvar.cpu(), rv, atol=2e-2)
    for a, b in zip(outs[True], outs[False]):          # the two forms agree to rounding of the 16-bit conv output
        assert rel_l2(a, b) < tol


@pytest.mark.parametrize("H,W,k,stride,pad", [(9, 11, 3, 2, 1), (12, 16, 3, 2, 1), (10, 10, 2, 2, 0), (7, 9, 3, 1, 1)])
def test_maxpool_fwd_bwd_matches_torch(dvt, device, H, W, k, stride, pad):
    """nn.MaxPool2d (custom_resnet.py:107 uses 3 / 2 / 1: the specialised backward; other geometries: the generic one),
    first-max tie rule, odd and even map sizes."""
    g = torch.Generator().manual_seed(33)
    N, C = 2, 16
    x = torch.relu(torch.randn(N, C, H, W, generator=g))          # many exact zeros -> ties
    xr = x.clone().requires_grad_(True)
    ref = TF.max_pool2d(xr, k, stride, pad)
    xd = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda().requires_grad_(True)
    y = dvt.functional.maxpool_nhwc(xd, N, C, H, W, k, stride, pad)
    assert torch.equal(y.cpu(), ref.permute(0, 2, 3, 1).reshape(-1, C))
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    y.backward(gy.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda())
    assert torch.allclose(xd.grad.cpu(), xr.grad.permute(0, 2, 3, 1).reshape(-1, C), atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,C", [(37, 8), (4099, 48), (70001, 64), (9000, 512), (30011, 144), (4099, 240), (5000, 464), (2744, 928), (1000, 296)])
def test_batchnorm_row_streaming_kernels_and_relu_mask(dvt, device, dtype, rows, C):
    """BatchNorm + residual + ReLU of a BasicBlock's second layer (custom_resnet.py:48-52) and its backward against plain
    fp64 arithmetic on the same rounded operands -- row counts that are not multiples of the kernels' row sweep, a channel
    group count (48 / 8 = 6) that does not divide the thread count --, and the ReLU mask bytes of the forward: bit k of
    byte (r, g) is y[r, 8g + k] > 0, and the backward fed with them is BIT-identical to the backward fed with y."""
    from dvt_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    z = torch.randn(rows, C, generator=g).to(dtype)
    res = torch.randn(rows, C, generator=g).to(dtype)
    dy = torch.randn(rows, C, generator=g).to(dtype)
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    zd, rd, dyd, gd, bd = (t.cuda() for t in (z, res, dy, gamma, beta))
    mean, invstd = ops.bn_stats(zd, None, None, 1e-5, 0.1)
    z64 = z.double()
    mu, var = z64.mean(0), z64.var(0, unbiased=False)
    assert torch.allclose(mean.cpu().double(), mu, atol=1e-4) and rel_l2(invstd.cpu().double(), (var + 1e-5).rsqrt()) < 1e-4
    y, mask = ops.bn_apply_fwd(zd, mean, invstd, gd, bd, rd, True, want_mask=True)
    y_plain = ops.bn_apply_fwd(zd, mean, invstd, gd, bd, rd, True)
    assert torch.equal(y, y_plain)
    xh = (z64 - mean.cpu().double()) * invstd.cpu().double()
    ref = torch.relu(xh * gamma.double() + beta.double() + res.double())
    tol = 1e-5 if dtype == torch.float32 else (6e-3 if dtype == torch.bfloat16 else 8e-4)
    assert rel_l2(y.float().cpu().double(), ref) < tol
    bits = (y.float() > 0).view(rows, C // 8, 8).to(torch.int32)
    want = (bits << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(mask, want)
    # backward: mask-fed == y-fed, bit for bit; and both against float64
    dz_m, dres_m, dg_m, db_m = ops.bn_bwd(dyd, zd, None, mean, invstd, gd, True, True, True, mask=mask)
    dz_y, dres_y, dg_y, db_y = ops.bn_bwd(dyd, zd, y, mean, invstd, gd, True, True, True)
    assert torch.equal(dz_m, dz_y) and torch.equal(dres_m, dres_y) and torch.equal(dg_m, dg_y) and torch.equal(db_m, db_y)
    on = (y.float().cpu() > 0).double()
    dzr = dy.double() * on
    dg_ref, db_ref = (dzr * xh).sum(0), dzr.sum(0)
    dx_ref = gamma.double() * invstd.cpu().double() * (dzr - db_ref / rows - xh * dg_ref / rows)
    assert rel_l2(dg_m.cpu().double(), dg_ref) < 20 * tol and rel_l2(db_m.cpu().double(), db_ref) < 20 * tol
    assert rel_l2(dz_m.float().cpu().double(), dx_ref) < tol and torch.equal(dres_m.float().cpu().double(), dzr)
    # a ReLU layer without a residual branch: the mask is recomputed from z with the forward's own formula
    y2 = ops.bn_apply_fwd(zd, mean, invstd, gd, bd, None, True)
    dz_r, _, dg_r, db_r = ops.bn_bwd(dyd, zd, None, mean, invstd, gd, True, True, False, beta=bd)
    dz_s, _, dg_s, db_s = ops.bn_bwd(dyd, zd, y2, mean, invstd, gd, True, True, False)
    # (identical unless an activation is positive in fp32 and rounds to zero in 16 bits: a handful of elements at most)
    assert rel_l2(dz_r.float(), dz_s.float()) < 1e-4 and rel_l2(dg_r, dg_s) < 1e-5 and rel_l2(db_r, db_s) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,W", [(9, 11), (12, 16)])
def test_bn_relu_maxpool_in_one_pass_matches_the_three_ops(dvt, device, dtype, H, W):
    """The stem's bn1 -> relu -> maxpool(3, 2, 1) (custom_resnet.py:100-105,138-142) as one forward pass and a BatchNorm
    backward that gathers its incoming gradient from the pooled gradient: same pooled values and argmax taps as the three
    separate kernels, same dz / dgamma / dbeta (the fused backward skips one 16-bit rounding of the un-pooled gradient)."""
    from dvt_amd import ops
    g = torch.Generator().manual_seed(35)
    N, C = 3, 16
    z = torch.randn(N * H * W, C, generator=g).to(dtype).cuda()
    mean = (0.2 * torch.randn(C, generator=g)).cuda()
    invstd = (0.5 + torch.rand(C, generator=g)).cuda()
    gamma = torch.randn(C, generator=g).cuda()                    # both signs: the affine map is not monotonic in z
    beta = (0.3 * torch.randn(C, generator=g)).cuda()
    y_ref = ops.bn_apply_fwd(z, mean, invstd, gamma, beta, None, True)
    p_ref, i_ref = ops.maxpool_fwd(y_ref, N, C, H, W, 3, 2, 1)
    p, i = ops.bn_relu_maxpool_fwd(z, mean, invstd, gamma, beta, N, C, H, W, True)
    assert torch.equal(p, p_ref) and torch.equal(i, i_ref)
    dyp = torch.randn(p.shape, generator=g).to(dtype).cuda()
    dx_ref = ops.maxpool_bwd(dyp, i_ref, N, C, H, W, 3, 2, 1)
    dz_ref, _, dg_ref, db_ref = ops.bn_bwd(dx_ref, z, None, mean, invstd, gamma, True, True, False, beta=beta)
    dz, dg, db = ops.bn_bwd_pooled(dyp, i, z, mean, invstd, gamma, beta, N, H, W, True, True)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel_l2(dz.float(), dz_ref.float()) < tol
    assert rel_l2(dg, dg_ref) < tol and rel_l2(db, db_ref) < tol
    # many ties (a map of zeros and ones under the ReLU): the first maximal tap of every window
    zt = (torch.rand(N * H * W, C, generator=g) < 0.3).to(dtype).cuda()
    one, zero = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    yt = ops.bn_apply_fwd(zt, zero, one, one, zero, None, True)
    pt_ref, it_ref = ops.maxpool_fwd(yt, N, C, H, W, 3, 2, 1)
    pt, it = ops.bn_relu_maxpool_fwd(zt, zero, one, one, zero, N, C, H, W, True)
    assert torch.equal(pt, pt_ref) and torch.equal(it, it_ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W", [(3, 56, 56), (2, 28, 28), (2, 13, 20), (1, 5, 56), (300, 8, 8)])
def test_conv3x3_c64_from_an_lds_halo_patch(dvt, device, dtype, N, H, W):
    """dvt_conv3x3_c64 (layer 1 of ResNet-18, custom_resnet.py:19-22,109: 64 -> 64 channels, 3 x 3 / 1 / 1) against
    F.conv2d on the same 16-bit operands, and its BatchNorm partial sums against the statistics of its own output; tile
    heights that do and do not divide H, more tiles than CUs (persistent loop, both patch buffers)."""
    from dvt_amd import ops
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, 64, H, W, generator=g).to(dtype)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05)
    ref = TF.conv2d(x.float(), w.to(dtype).float(), padding=1).permute(0, 2, 3, 1).reshape(-1, 64)
    xd = x.permute(0, 2, 3, 1).reshape(-1, 64).contiguous().cuda()
    wp = ops.conv_weight_pack(w.cuda(), 576, dtype)
    assert ops.conv3x3_c64_supported(xd, wp, N, H, W)
    z, partial, parts = ops.conv3x3_c64(xd, wp, N, H, W, want_stats=True)
    assert rel_l2(z.float().cpu(), ref) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    mean, invstd = ops.bn_stats_from_partials(partial, parts, z.shape[0], 64, None, None, 1e-5, 0.1)
    zf = ref.double()
    assert torch.allclose(mean.cpu().double(), zf.mean(0), atol=2e-3)
    assert rel_l2(invstd.cpu().double(), 1.0 / torch.sqrt(zf.var(0, unbiased=False) + 1e-5)) < 2e-3
    z2 = ops.conv3x3_c64(xd, wp, N, H, W)                            # no statistics: same values
    assert torch.equal(z, z2)
    res = torch.randn(N * H * W, 64, generator=g).to(dtype).cuda()     # the shortcut's gradient joining a data gradient
    z3 = ops.conv3x3_c64(xd, wp, N, H, W, residual=res)
    assert torch.equal(z3, (z.float() + res.float()).to(dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Cin,Cout", [(64, 144), (144, 64), (128, 288), (288, 128)])
@pytest.mark.parametrize("N,H,W", [(3, 56, 56), (2, 28, 28), (2, 13, 20), (1, 5, 56), (300, 8, 8), (150, 9, 40)])
def test_conv3x3_stream_halo_patch_with_streamed_weights(dvt, device, dtype, Cin, Cout, N, H, W):
    """dvt_conv3x3_stream (the spatial half of R(2+1)D-18's layer-1 Conv2Plus1D, video_resnet.py: 64 -> 144 forward and
    144 -> 64 as the data gradient) against F.conv2d on the same 16-bit operands; BatchNorm partial sums of the 144-wide
    output against the statistics of that output, the residual of the 64-wide one; tile heights that do and do not divide
    H, fewer and (many) more tiles than CUs (producer ring over several tiles, both patch buffers, clamped last tile)."""
    from dvt_amd import ops
    g = torch.Generator().manual_seed(43)
    x = torch.randn(N, Cin, H, W, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05)
    ref = TF.conv2d(x.float(), w.to(dtype).float(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    xd = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda()
    wp = ops.conv_weight_pack(w.cuda(), 9 * Cin, dtype)
    assert ops.conv3x3_stream_supported(xd, wp, N, H, W, Cin, Cout)
    tol = 4e-3 if dtype == torch.bfloat16 else 6e-4
    if Cout % 144 == 0:
        z, partial, parts = ops.conv3x3_stream(xd, wp, N, H, W, Cin, Cout, want_stats=True)
        assert rel_l2(z.float().cpu(), ref) < tol
        mean, invstd = ops.bn_stats_from_partials(partial, parts, z.shape[0], Cout, None, None, 1e-5, 0.1)
        zf = ref.double()
        assert torch.allclose(mean.cpu().double(), zf.mean(0), atol=2e-3)
        assert rel_l2(invstd.cpu().double(), 1.0 / torch.sqrt(zf.var(0, unbiased=False) + 1e-5)) < 2e-3
        assert torch.equal(z, ops.conv3x3_stream(xd, wp, N, H, W, Cin, Cout))     # no statistics: same values
    else:
        z = ops.conv3x3_stream(xd, wp, N, H, W, Cin, Cout)
        assert rel_l2(z.float().cpu(), ref) < tol
        res = torch.randn(N * H * W, Cout, generator=g).to(dtype).cuda()          # the shortcut's gradient joining in
        z3 = ops.conv3x3_stream(xd, wp, N, H, W, Cin, Cout, residual=res)
        assert torch.equal(z3, (z.float() + res.float()).to(dtype))
    worst = (z.float().cpu() - ref).abs().max().item()
    assert worst < (0.25 if dtype == torch.bfloat16 else 0.03), worst           # no stray pixel hidden inside the L2 norm


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,T,H,W", [(3, 12, 56, 56), (2, 5, 14, 8), (30, 7, 8, 8), (1, 16, 28, 28)])
def test_conv3x1_stream_temporal_data_gradient(dvt, device, dtype, N, T, H, W):
    """dvt_conv3x1_stream (the temporal half of R(2+1)D-18's layer-1 Conv2Plus1D as its data gradient: 64 -> 144 channels over
    the [T, H*W] view of a clip, rows H*W pixels apart) against F.conv2d with a (3, 1) filter on the same 16-bit operands;
    frame counts that the tile height does and does not divide, more tiles than CUs."""
    from dvt_amd import ops
    g = torch.Generator().manual_seed(47)
    HW = H * W
    x = torch.randn(N, 64, T, HW, generator=g).to(dtype)
    w = (torch.randn(144, 64, 3, 1, generator=g) * 0.07)
    ref = TF.conv2d(x.float(), w.to(dtype).float(), padding=(1, 0)).permute(0, 2, 3, 1).reshape(-1, 144)
    xd = x.permute(0, 2, 3, 1).reshape(-1, 64).contiguous().cuda()
    wp = ops.conv_weight_pack(w.cuda(), 3 * 64, dtype)
    assert ops.conv3x1_stream_supported(xd, wp, N, T, HW, 64, 144)
    z = ops.conv3x1_stream(xd, wp, N, T, HW, 64, 144)
    assert rel_l2(z.float().cpu(), ref) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    assert (z.float().cpu() - ref).abs().max().item() < (0.25 if dtype == torch.bfloat16 else 0.03)


def test_maxpool_first_max_and_eval_bn(dvt, device):
    g = torch.Generator().manual_seed(32)
    N, C, H, W = 2, 8, 9, 11
    x = torch.relu(torch.randn(N, C, H, W, generator=g))          # many exact zeros -> ties
    xr = x.clone().requires_grad_(True)
    ref = TF.max_pool2d(xr, 3, 2, 1)
    xd = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda().requires_grad_(True)
    y = dvt.functional.maxpool_nhwc(xd, N, C, H, W, 3, 2, 1)
    assert torch.equal(y.cpu(), ref.permute(0, 2, 3, 1).reshape(-1, C))
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    y.backward(gy.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda())
    assert torch.allclose(xd.grad.cpu(), xr.grad.permute(0, 2, 3, 1).reshape(-1, C), atol=1e-6)
    # eval-mode BatchNorm uses the running statistics
    conv = torch.nn.Conv2d(8, 8, 1, bias=False)
    bn = torch.nn.BatchNorm2d(8).eval()
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(8, generator=g)); bn.running_var.copy_(torch.rand(8, generator=g) + 0.5)
    refe = torch.relu(bn(conv(x)))
    ye = dvt.functional.conv_bn_act(xd.detach(), conv.cuda(), bn.cuda(), (N, 8, H, W, False), relu=True,
                                    dtype=torch.float32)
    assert rel_l2(ye, refe.permute(0, 2, 3, 1).reshape(-1, 8)) < 1e-5


# fp32 gradient bound 1e-2: one ReLU-mask flip of an activation that is 0 +- 1 ulp on the 98 x 512 layer-4 map moves
# every upstream gradient by 1/sqrt(50176) = 4.5e-3 (seen after a change of the BatchNorm summation order); 1e-4 otherwise.
# 16-bit kernels: this fixture (17 BatchNorm'd ReLU layers on batch statistics, random output gradients) is
# ill-conditioned in a short mantissa WHOEVER computes it -- the reference's own torch.autocast(bf16) CPU run deviates
# from its fp32 run by 1.7e-2 / 2.9e-2 / 4.6e-2 on (x2, x3, x4) and by 0.10 .. 0.45 on the parameter gradients (cosines
# down to 0.90), at batch 2 and at batch 16 alike; fp16: 6e-3 on x4, 0.03 .. 0.14 on gradients
# (tests/golden/resnet18_lowprec.npz, written by tools/gen_golden.py from the imported reference).  The HIP path is
# held to 1.5x the reference's own deviation, per output and per parameter.
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["fp32", "bf16", "fp16"])
def test_resnet18_pyramid_matches_reference_golden(dvt, device, dtype):
    """custom_resnet.resnet18 at 224x224, train mode, vs the imported reference."""
    from dvt_amd.models.custom_resnet import resnet18
    g = golden("resnet18_pyramid.npz")
    lp = golden("resnet18_lowprec.npz")
    tag = {torch.bfloat16: "bf16", torch.float16: "fp16"}.get(dtype)
    lp_names = [str(n) for n in lp["names"]]
    net = resnet18(False, compute_dtype=dtype)
    rng = np.random.default_rng(int(g["seed"]))
    fill_resnet_from_numpy(net, rng)
    x = torch.from_numpy(rng.standard_normal((2, 3, 224, 224)).astype(np.float32))
    net = net.cuda().train()
    x2, x3, x4 = net(x.cuda())
    assert x2.shape == (2, 128, 28, 28) and x3.shape == (2, 256, 14, 14) and x4.shape == (2, 512, 7, 7)
    errs = [rel_l2(t, T(g[k])) for t, k in ((x2, "x2"), (x3, "x3"), (x4, "x4"))]
    print(f"[resnet18/{dtype}] pyramid rel errors {errs}" + (f" (reference's own {list(lp[tag + ':out_err'])})" if tag else ""))
    for i, e in enumerate(errs):
        assert e < (2e-4 if tag is None else 1.5 * float(lp[tag + ":out_err"][i])), (i, e)
    gs = [torch.from_numpy(rng.standard_normal(tuple(t.shape)).astype(np.float32)) for t in (x2, x3, x4)]
    # the scalar of the fixture: sum_i <x_i, g_i> / 1000  -> gradient g_i / 1000 on each output
    # fp16: the fixture's 1e-3-scaled output gradients underflow half precision on the way down -> static loss scale
    ls = 1024.0 if dtype == torch.float16 else 1.0
    torch.autograd.backward([x2, x3, x4], [(gg * (ls / 1000.0)).to(t.dtype).cuda() for t, gg in zip((x2, x3, x4), gs)])
    P = dict(net.named_parameters())
    for p_ in P.values():
        if p_.grad is not None:
            p_.grad.div_(ls)
    worst, worst_ratio = 0.0, 0.0
    for k in g.files:
        if k.startswith("g:"):
            name = k[2:]
            e = rel_l2(P[name].grad, T(g[k]))
            a, b = P[name].grad.double().cpu().reshape(-1), T(g[k]).double().reshape(-1)
            cos = float(a @ b / (a.norm() * b.norm()))
            worst = max(worst, e)
            if tag is None:
                assert e < 1e-2 and cos > 0.9998, (k, e, cos)
            else:
                i = lp_names.index(name)
                ref_e, ref_cos = float(lp[tag + ":grad_err"][i]), float(lp[tag + ":grad_cos"][i])
                worst_ratio = max(worst_ratio, e / ref_e)
                assert e < 1.5 * ref_e, (k, e, ref_e)
                assert 1.0 - cos < 2.0 * (1.0 - ref_cos), (k, cos, ref_cos)
    names, norms = list(g["grad_names"]), g["grad_norms"]
    for n, ref_norm in zip(names, norms):
        got = float(P[str(n)].grad.double().norm())
        slack = 0.02 if tag is None else max(0.05, 1.5 * float(lp[tag + ":grad_err"][lp_names.index(str(n))]))
        assert abs(got - ref_norm) <= slack * ref_norm + 1e-6, (n, got, ref_norm)
    print(f"[resnet18/{dtype}] worst stored-grad rel {worst:.2e}" + (f", worst ours/reference's-own ratio {worst_ratio:.2f}" if tag else ""))
    assert torch.allclose(net.bn1.running_mean.cpu(), T(g["rm:bn1"]), atol=1e-4 if dtype == torch.float32 else 2e-2)
    assert P["fc.weight"].grad is None          # the reference's avgpool+fc tail is dead code


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_tpn_pyramid_and_reasoning_match_reference(dvt, device, dtype, tol):
    """pyramid pooling + 1x1 conv, sum_group, Reasoning MLPs vs the golden produced
    by executing the reference's TPN.py."""
    from dvt_amd.models.TPN import Reasoning, Feature_Pyramid_low, Feature_Pyramid_Mid, Feature_Pyramid_High, sum_group
    g = golden("tpn_pieces.npz")
    rng = np.random.default_rng(int(g["seed"]))
    reason = Reasoning()
    fill_resnet_from_numpy(reason, rng)
    x = torch.from_numpy(rng.standard_normal((1, 20, 896)).astype(np.float32))
    reason = reason.cuda().eval()
    xd = x.to(dtype).cuda().requires_grad_(True)
    y = reason(xd)
    assert y.shape == (1, 15) and rel_l2(y, T(g["reason_out"])) < tol
    gy = torch.from_numpy(rng.standard_normal((1, 15)).astype(np.float32))
    y.backward(gy.cuda())
    # bf16: ReLU masks of near-zero group sums flip under rounding -> input-gradient noise
    assert rel_l2(xd.grad, T(g["reason_gx"])) < (3 * tol if dtype == torch.float32 else 0.15)
    assert rel_l2(reason.relation[2][7].weight.grad, T(g["reason_gw_last"])) < 3 * tol
    assert rel_l2(sum_group(x.to(dtype).cuda(), 3), T(g["sum_group3"])) < tol
    yt = reason.train()(xd)                      # Dropout(0.6) / Dropout(0.5) active: a different, finite prediction
    assert yt.shape == (1, 15) and torch.isfinite(yt).all() and not torch.equal(yt, y.detach())
    reason.eval()
    for name, cls, shape in (("low", Feature_Pyramid_low, (3, 128, 28, 28)), ("mid", Feature_Pyramid_Mid, (3, 256, 14, 14)),
                             ("high", Feature_Pyramid_High, (3, 512, 7, 7))):
        m = cls()
        fill_resnet_from_numpy(m, rng)
        f = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
        N, C, H, W = shape
        fm = (f.to(dtype).permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda(), N, H, W)
        assert rel_l2(m.cuda()(fm), T(g["pyr_" + name])) < tol, name


def test_tpn_end_to_end_tokens(dvt, device):
    """TPN.frame_tokens on 20 frames (fp32 mode) vs the oracle composition resnet34 -> pyramid -> cat."""
    from oracle import cnn_path as C
    from dvt_amd.models.TPN import TPN
    torch.manual_seed(5)
    net = TPN(compute_dtype=torch.float32)
    rng = np.random.default_rng(77)
    fill_resnet_from_numpy(net, rng)
    x = torch.from_numpy(rng.standard_normal((4, 3, 224, 224)).astype(np.float32))
    P = {k: v.detach().clone() for k, v in net.net.state_dict().items()}
    with torch.no_grad():
        x2, x3, x4, _ = C.resnet_pyramid(x, P, [3, 4, 6, 3], True)
        sd = net.state_dict()
        ref = torch.cat((C.pyramid_vector(x4),
                         C.pyramid_vector(x3, sd["pyramid_mid.channels_reduce.weight"], sd["pyramid_mid.channels_reduce.bias"]),
                         C.pyramid_vector(x2, sd["pyramid_low.channels_reduce.weight"], sd["pyramid_low.channels_reduce.bias"])), dim=-1)
    net = net.cuda().train()
    tok = net.frame_tokens(x.cuda())
    assert tok.shape == (4, 896) and rel_l2(tok, ref) < 2e-4


@pytest.mark.parametrize("mode", ["fp32", "bf16"])       # (torch's CPU conv3d in fp16 takes ~25 s per oracle run)
def test_r2plus1d_18_matches_conv3d_restatement(dvt, device, mode):
    """R(2+1)D-18 features (factorised (1,3,3)+(3,1,1) convolutions, BatchNorm3d -- batch statistics in
    the fp32 case --, strided 1x1x1 downsample), forward and EVERY parameter gradient, vs a torch-CPU conv3d restatement.  UNPINNED against
    torchvision (not installed).  Protocol (tests/util.py): fp32 kernels against the restatement in float64, within 2x
    the deviation of its own fp32 run from that (ReLU-mask flips set an fp32 gradient noise floor of ~1e-2 behind 20
    BatchNorm'd ReLU layers, whoever computes them); 16-bit kernels against its fp32 run, within 2x the larger of its
    own autocast and cast-to-16-bit deviations on the same inputs."""
    from oracle import cnn_path as C
    from dvt_amd.models.video_resnet import r2plus1d_18
    dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[mode]
    net = r2plus1d_18(False, compute_dtype=dtype)
    rng = np.random.default_rng(91)
    with torch.no_grad():
        for name, p in net.named_parameters():
            a = rng.standard_normal(tuple(p.shape)).astype(np.float32)
            if p.dim() == 5:
                a *= np.float32(np.sqrt(2.0 / (p.shape[1] * p.shape[2] * p.shape[3] * p.shape[4])))
            elif p.dim() == 2:
                a *= np.float32(0.02)
            elif name.endswith("weight"):
                a = 1 + np.float32(0.1) * a
            else:
                a = np.float32(0.1) * a
            p.copy_(torch.from_numpy(a))
    x = torch.from_numpy(rng.standard_normal((4, 3, 8, 48, 48)).astype(np.float32))     # deepest BatchNorm: 36 samples
    gy = torch.from_numpy(rng.standard_normal((4, 512)).astype(np.float32))
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    pnames = [k for k, _ in net.named_parameters() if not k.startswith("fc.")]

    training = mode == "fp32"      # 16-bit: BatchNorm on running statistics -- on batch statistics the restatement's own bf16 gradients
    # deviate from its fp32 ones by ~100 % at any size that runs on the CPU in seconds (36 samples per channel in layer 4)

    def oracle(kind, dt):
        cast = (lambda t: t.to(dt)) if kind == "cast" else (lambda t: t)
        P = {k: (cast(v) if v.dtype.is_floating_point else v).clone() for k, v in state.items()}
        for k in pnames:
            P[k].requires_grad_(True)
        scale = 256.0 if dt == torch.float16 else 1.0
        if kind == "amp":
            with torch.autocast("cpu", dtype=dt):
                out = C.r2plus1d_features(x, P, training=training)
        else:
            out = C.r2plus1d_features(cast(x), P, training=training)
        (out.double() * gy.double()).sum().mul(scale).backward()
        return out.detach().double(), {k: P[k].grad.double() / scale for k in pnames}

    if mode == "fp32":
        truth, tg = oracle("cast", torch.float64)
        o, og = oracle("cast", torch.float32)
        yard_out, yard = rel_l2(o, truth), {k: rel_l2(og[k], tg[k]) for k in tg}
    else:
        truth, tg = oracle("cast", torch.float32)
        yard_out, yard = 0.0, {k: 0.0 for k in tg}
        for kind in ("amp", "cast"):
            o, og = oracle(kind, dtype)
            yard_out = max(yard_out, rel_l2(o, truth))
            yard = {k: max(yard[k], rel_l2(og[k], tg[k])) for k in tg}
    net = net.cuda().train(training)
    out = net.features(x.cuda())
    assert out.shape == (4, 512)
    scale = 256.0 if mode == "fp16" else 1.0
    out.backward((gy * scale).to(out.dtype).cuda())
    Pn = dict(net.named_parameters())
    errs = {k: rel_l2(Pn[k].grad.double() / scale, tg[k]) for k in tg}
    e_out = rel_l2(out, truth)
    w = assert_within_reference_lowprec(f"r2plus1d/{mode}", e_out, errs, yard_out, yard,
                                        floor=1e-4 if mode == "fp32" else 2e-4, flip_noise=True)
    print(f"[r2plus1d/{mode}] features rel {e_out:.2e} (restatement's own {yard_out:.2e}); {len(errs)} gradients: median "
          f"{float(np.median(list(errs.values()))):.2e} (own {float(np.median(list(yard.values()))):.2e}); worst ratio "
          f"{w[0][1]:.2f} ({w[0][0]})")


def test_frame_transformer_with_reference_encoders(dvt, device):
    """The default FrameTransformer (VidResNet R(2+1)D-18 encoder, 14 tokens x 896, 19 classes) runs a
    training step end to end on [B, 13, 12, 3, 112, 112] chunks (batch contract of MMX_Light_dl.py:286)."""
    from dvt_amd.models.frame_transformer import FrameTransformer
    torch.manual_seed(1)
    net = FrameTransformer(batch_size=1, seq_len=13, cls=1, model="vid", opt="adamW", learning_rate=5e-6,
                           weight_decay=0.09, momentum=0.005, frame_len=4).cuda().eval()   # 4-frame chunks keep it light
    g = torch.Generator().manual_seed(2)
    vid = torch.randn(1, 13, 4, 3, 112, 112, generator=g).cuda()
    target = (torch.rand(1, 19, generator=g) < 0.3).float().cuda()
    loss = net.training_step((target, None, vid), 0)
    loss.backward()
    assert torch.isfinite(loss.detach()).item()
    assert net.vid_cls.grad is not None and torch.isfinite(net.vid_cls.grad).all()
    assert torch.isfinite(net.vid_model.backbone.stem[0].weight.grad).all()
    # the stem computes the pixel gradient for the CLS chunks only; the full computation gives the same CLS gradient
    g_restricted = net.vid_cls.grad.clone()
    w_restricted = net.vid_model.backbone.stem[0].weight.grad.clone()
    net.zero_grad()
    net.restrict_pixel_grad = False
    net.training_step((target, None, vid), 0).backward()
    assert float(g_restricted.abs().max()) > 0
    assert torch.equal(net.vid_cls.grad, g_restricted) and torch.equal(net.vid_model.backbone.stem[0].weight.grad, w_restricted)


def test_default_frame_transformer_vid_step_matches_oracle(dvt, device):
    """The reference-DEFAULT composition (frame_transformer.py:192-210 with the encoder of :64-74): CLS chunk + chunks
    -> R(2+1)D-18 (TRAIN-mode BatchNorm3d) -> fc(512 -> 896) -> positional table -> 4 post-norm encoder layers -> CLS
    -> 3-layer head -> BCE, fp32 kernels, forward AND backward, every parameter gradient, against
    oracle.cnn_path.r2plus1d_features + oracle.clip_path.{transformer_base, mlp_head3} on the same weights.
    Dropout is set to 0 (torch's mask stream cannot be reproduced); chunks are 8 x 64^2 so that the deepest BatchNorm
    sees 128 samples per channel.  The R(2+1)D stage itself stays unpinned against torchvision."""
    from oracle import cnn_path as C
    from oracle import clip_path as O
    from dvt_amd.models.frame_transformer import FrameTransformer
    torch.manual_seed(3)
    net = FrameTransformer(batch_size=2, seq_len=3, cls=1, model="vid", opt="adamW", learning_rate=5e-6, weight_decay=0.09,
                           momentum=0.005, frame_len=8, clip_size=64, tokens=4, encoder_dropout=0.0,
                           compute_dtype=torch.float32)
    rng = np.random.default_rng(93)
    with torch.no_grad():
        for name, p in net.named_parameters():
            a = rng.standard_normal(tuple(p.shape)).astype(np.float32)
            if name == "vid_cls":
                a = np.float32(0.5) * a
            elif p.dim() == 5:
                a *= np.float32(np.sqrt(2.0 / (p.shape[1] * p.shape[2] * p.shape[3] * p.shape[4])))
            elif p.dim() == 2:
                a *= np.float32(1.0 / np.sqrt(p.shape[1]))
            elif name.endswith("weight"):
                a = 1 + np.float32(0.1) * a
            else:
                a = np.float32(0.1) * a
            p.copy_(torch.from_numpy(a))
    B = 2
    vid = torch.from_numpy(rng.standard_normal((B, 3, 8, 3, 64, 64)).astype(np.float32))
    target = torch.from_numpy((rng.random((B, 19)) < 0.3).astype(np.float32))
    state = {k: v.detach().clone() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    pnames = [k for k, _ in net.named_parameters()]

    def oracle(dt):
        """vid_step (:192-210) + head + BCE on explicit formulas, in ``dt``: (logits, loss, {name: gradient})."""
        P = {k: v.to(dt).clone() for k, v in state.items()}
        for k in pnames:
            P[k].requires_grad_(True)
        cls = P["vid_cls"]                                                            # [1, T, 3, H, W]
        data = torch.cat((cls.unsqueeze(0).expand(B, *cls.shape), vid.to(dt)), dim=1)  # [B, 4, T, 3, H, W]   (:194-196)
        data = data.reshape(-1, *data.shape[2:]).permute(0, 2, 1, 3, 4)               # [B*4, 3, T, H, W]    (:197-198)
        bbP = {k[len("vid_model.backbone."):]: v for k, v in P.items() if k.startswith("vid_model.backbone.")}
        feats = C.r2plus1d_features(data, bbP, training=True)                         # [B*4, 512]
        emb = O.linear(feats, P["vid_model.backbone.fc.0.weight"], P["vid_model.backbone.fc.0.bias"])
        seq = emb.reshape(B, 4, -1).permute(1, 0, 2) + P["position_encoder.pe"][:4]   # (:203-206)
        logits = O.mlp_head3(O.transformer_base(seq, P, "distil_transformer.", 4, 2)[0], P)
        loss = O.bce_with_logits(logits, target.to(dt))
        loss.backward()
        return logits.detach(), loss.detach(), {k: P[k].grad for k in pnames if P[k].grad is not None}

    # Truth = the oracle in float64; yardstick = the oracle's own fp32 run against it.  Behind 20 BatchNorm'd ReLU layers
    # the fp32 noise floor of a gradient is set by ReLU-mask flips of activations that are 0 +- round-off: torch's own
    # fp32 gradients deviate from its float64 ones by ~8e-3 (median) here, whoever computes them.
    l64, loss64, g64 = oracle(torch.float64)
    l32, loss32, g32 = oracle(torch.float32)
    own = {k: rel_l2(g32[k], g64[k]) for k in g64}
    net = net.cuda().train()                                                      # BatchNorm on batch statistics
    loss = net.training_step((target.cuda(), None, vid.cuda()), 0)
    loss.backward()
    e_loss = abs(float(loss.detach()) - float(loss64))
    Pn = dict(net.named_parameters())
    errs = {k: rel_l2(Pn[k].grad, g64[k]) for k in g64 if Pn[k].grad is not None}
    missing = [k for k in pnames if (k in g64) != (Pn[k].grad is not None) and k not in ("norm.weight", "norm.bias")]
    assert not missing, missing
    wk = max(errs, key=lambda k: errs[k] / (own[k] + 1e-4))
    med, med_own = float(np.median(list(errs.values()))), float(np.median(list(own.values())))
    print(f"[default FrameTransformer vid/fp32] loss abs {e_loss:.2e}; {len(errs)} gradients vs float64: median {med:.2e} "
          f"(oracle's own fp32: {med_own:.2e}); worst ratio {wk} {errs[wk]:.2e} vs {own[wk]:.2e}")
    assert e_loss < 1e-5 and len(errs) > 100
    assert med <= 2 * med_own + 1e-4
    for k, e in errs.items():
        assert e <= 2 * max(own[k], med_own) + 1e-4, (k, e, own[k])
    with torch.no_grad():
        logits = net(None, vid.cuda())
    assert rel_l2(logits, l64) < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [
    # N, Cin, H, W, Cout, k, stride, pad
    (3, 64, 20, 20, 64, 3, 1, 1),          # layer1: cfg 256x128x32, ragged rows
    (2, 64, 18, 22, 128, 3, 2, 1),         # strided 3x3
    (2, 128, 14, 14, 256, 3, 1, 1),        # cfg 256x256x64
    (2, 256, 9, 11, 512, (3, 1), 1, (1, 0)),   # rectangular (temporal) kernel
    (4, 64, 16, 16, 128, 1, 2, 0),         # 1x1 stride-2 downsample
    (1, 512, 7, 7, 512, 3, 1, 1),
    (16, 64, 56, 56, 144, 3, 1, 1),        # 144 output columns over many rows (R(2+1)D-18 layer 1)
])
def test_implicit_gemm_convolution_matches_explicit_path(dvt, device, dtype, geom):
    """dvt_conv2d_implicit (gather fused into the GEMM operand DMA) == dvt_im2col + dvt_gemm, forward and the
    stride-1 data gradient (rotated weights); same MFMA products in the same k order, so the match is tight."""
    ops = dvt.ops
    N, Cin, H, W, Cout, k, stride, pad = geom
    g = torch.Generator().manual_seed(123)
    (kh, kw) = ops._pair(k)
    x = torch.randn(N * H * W, Cin, generator=g).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).cuda()
    K = kh * kw * Cin
    wp = ops.conv_weight_pack(w, K, dtype)
    assert ops.conv2d_implicit_supported(x, wp, N, Cin, H, W, Cout, k, stride, pad)
    y = ops.conv2d_implicit(x, wp, N, Cin, H, W, Cout, k, stride, pad)
    col = ops.im2col(x, False, N, Cin, H, W, k, stride, pad, K, dtype)
    ref = ops.linear_fwd(col, wp)
    assert y.shape == ref.shape and rel_l2(y, ref) < 1e-3
    xr = x.float().cpu().view(N, H, W, Cin).permute(0, 3, 1, 2)
    cpu = torch.nn.functional.conv2d(xr, w.cpu(), None, stride, pad).permute(0, 2, 3, 1).reshape(-1, Cout)
    assert rel_l2(y, cpu) < (1e-2 if dtype == torch.bfloat16 else 2e-3)
    Ho, Wo = ops.conv_out_hw(H, W, k, stride, pad)
    (ph, pw) = ops._pair(pad)
    wd = ops.conv_weight_pack_dgrad(w, dtype)
    dz = torch.randn(N * Ho * Wo, Cout, generator=g).to(dtype).cuda()
    if ops._pair(stride) == (1, 1) and ops.conv2d_implicit_supported(dz, wd, N, Cout, Ho, Wo, Cin, k, 1, (kh - 1 - ph, kw - 1 - pw)):
        dx = ops.conv2d_implicit(dz, wd, N, Cout, Ho, Wo, Cin, k, 1, (kh - 1 - ph, kw - 1 - pw))
        dref = ops.col2im(ops.linear_dgrad(dz, wp), N, Cin, H, W, k, stride, pad)
        assert dx.shape == dref.shape and rel_l2(dx, dref) < (2e-2 if dtype == torch.bfloat16 else 3e-3)
        # a second gradient path (the block's shortcut) added on the accumulators: == the plain result + the operand, to one
        # rounding of the element type; the same operand through col2im's second input
        res = torch.randn(N * H * W, Cin, generator=g).to(dtype).cuda()
        dx_r = ops.conv2d_implicit(dz, wd, N, Cout, Ho, Wo, Cin, k, 1, (kh - 1 - ph, kw - 1 - pw), residual=res)
        eps16 = 8e-3 if dtype == torch.bfloat16 else 1e-3
        assert float((dx_r.float() - (dx.float() + res.float())).abs().max()) <= eps16 * float((dx.float().abs() + res.float().abs()).max())
        dref_r = ops.col2im(ops.linear_dgrad(dz, wp), N, Cin, H, W, k, stride, pad, add=res)
        assert float((dref_r.float() - (dref.float() + res.float())).abs().max()) <= eps16 * float((dref.float().abs() + res.float().abs()).max())
    # unsupported geometry is reported, not mis-computed
    x45 = torch.zeros(4 * 8 * 8, 40, dtype=dtype, device="cuda")
    assert not ops.conv2d_implicit_supported(x45, ops.conv_weight_pack(torch.zeros(64, 40, 3, 3, device="cuda"), 360, dtype),
                                             4, 40, 8, 8, 64, 3, 1, 1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [(8, 256, 14, 14, 512, 3, 1, 1),               # 1,568 rows x K = 2,304: 52 tiles of 128 x 128
                                  (56, 512, 7, 7, 1152, 3, 1, 1),               # R(2+1)D layer 4: 2,744 rows, 1,152 mid planes, K = 4,608
                                  (56, 1152, 2, 49, 512, (3, 1), 1, (1, 0)),    # ... its temporal half over the [T, H*W] view, K = 3,456
                                  (3, 1152, 7, 7, 512, 3, 1, 1)])               # 147 rows (a ragged row tile), K = 10,368: eight slices
def test_implicit_gemm_splits_a_deep_reduction_over_workgroups(dvt, device, dtype, geom):
    """dvt_conv2d_implicit_workspace_bytes > 0: launches that leave CUs empty and run a deep reduction (the forward / data
    gradient convolutions of R(2+1)D-18's layers 3 - 4, video_resnet.py:147-157) split K over blockIdx.z into fp32 slabs and sum
    them in a second launch that rounds to the map's type, adds the shortcut's gradient and leaves the BatchNorm partial sums:
    against conv2d in fp32 on the same rounded operands; statistics against the fp32 result; residual; a carried weight-gradient
    reduce handed to such a launch is performed (as a launch of its own) before it."""
    ops = dvt.ops
    import ctypes as C
    N, Cin, H, W, Cout, k, stride, pad = geom
    g = torch.Generator().manual_seed(N + Cin)
    (kh, kw) = ops._pair(k)
    x = torch.randn(N * H * W, Cin, generator=g).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).cuda()
    wp = ops.conv_weight_pack(w, kh * kw * Cin, dtype)
    d = ops._conv_desc(x, wp, None, N, Cin, H, W, Cout, k, stride, pad)
    lib = dvt._lib.load()
    assert lib.dvt_conv2d_implicit_workspace_bytes(C.byref(d)) > 0               # the split path is the one under test
    y, partial, parts = ops.conv2d_implicit(x, wp, N, Cin, H, W, Cout, k, stride, pad, want_stats=True)
    xr = x.float().view(N, H, W, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xr, w.to(dtype).float(), None, stride, pad).permute(0, 2, 3, 1).reshape(-1, Cout)   # (torch on the GPU: fp32 oracle of the same operands)
    tol = 5e-3 if dtype == torch.bfloat16 else 8e-4
    assert y.shape == ref.shape and rel_l2(y, ref) < tol
    assert parts == (y.shape[0] + 63) // 64
    mean, invstd = ops.bn_stats_from_partials(partial, parts, y.shape[0], Cout, None, None, 1e-5, 0.1)
    assert torch.allclose(mean, ref.mean(0), atol=2e-3) and rel_l2(invstd, (ref.var(0, unbiased=False) + 1e-5).rsqrt()) < 2e-3
    y2 = ops.conv2d_implicit(x, wp, N, Cin, H, W, Cout, k, stride, pad)         # without statistics: the same values
    assert torch.equal(y2, y)
    res = torch.randn(y.shape, generator=g).to(dtype).cuda()
    yr = ops.conv2d_implicit(x, wp, N, Cin, H, W, Cout, k, stride, pad, residual=res)
    assert rel_l2(yr, ref + res.float()) < tol
    # a pending split-K reduce (a weight gradient's) handed over as `carry`: done when the launch returns
    a = torch.randn(4096, 64, generator=g).to(dtype).cuda()
    b = torch.randn(4096, 128, generator=g).to(dtype).cuda()
    dw, pend = ops.linear_wgrad(a, b, defer_reduce=True)
    if pend is not None and pend.valid:
        y3 = ops.conv2d_implicit(x, wp, N, Cin, H, W, Cout, k, stride, pad, carry=pend)
        assert torch.equal(y3, y) and not pend.valid
        assert rel_l2(dw, a.float().t() @ b.float()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [(2, 144, 12, 49, 64, (3, 1), 1, (1, 0)),      # R(2+1)D-18 layer-1 temporal conv (un-padded mid planes)
                                  (2, 144, 12, 49, 128, (3, 1), (2, 1), (1, 0)),
                                  (3, 144, 14, 14, 256, 3, 1, 1),                # 64-deep k-tiles: 1296 -> 1344
                                  (2, 40, 17, 13, 64, 3, 2, 1), (1, 72, 9, 9, 200, (1, 3), 1, (0, 1)),
                                  (2, 8 * 29, 6, 6, 64, 3, 1, 1)])
def test_implicit_gemm_with_taps_that_split_a_k_tile(dvt, device, dtype, geom):
    """dvt_conv2d_implicit with C % 8 == 0 but no whole number of k-tiles per filter tap (the 144 mid planes of
    video_resnet.py:69 left un-padded): every lane of the operand DMA derives the tap of its own chunk, K is rounded up to the
    k-tile with zero weight columns.  Against conv2d on the same rounded operands; operands in front of NaN-filled memory."""
    ops = dvt.ops
    N, Cin, H, W, Cout, k, stride, pad = geom
    g = torch.Generator().manual_seed(77)
    (kh, kw) = ops._pair(k)
    xb = torch.full((N * H * W + 64, Cin), float("nan"), dtype=dtype, device="cuda")
    x = xb[: N * H * W]
    x.copy_(torch.randn(N * H * W, Cin, generator=g).to(dtype))
    w = (torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).cuda()
    ld = ops.conv2d_implicit_k(Cin, Cout, k)
    assert ld >= kh * kw * Cin and ld % 32 == 0
    wp = ops.conv_weight_pack(w, ld, dtype)
    assert ops.conv2d_implicit_supported(x, wp, N, Cin, H, W, Cout, k, stride, pad)
    y, partial, parts = ops.conv2d_implicit(x, wp, N, Cin, H, W, Cout, k, stride, pad, want_stats=True)
    xr = x.float().cpu().view(N, H, W, Cin).permute(0, 3, 1, 2)
    cpu = torch.nn.functional.conv2d(xr, w.to(dtype).float().cpu(), None, stride, pad).permute(0, 2, 3, 1).reshape(-1, Cout)
    assert y.shape == cpu.shape and rel_l2(y, cpu) < (5e-3 if dtype == torch.bfloat16 else 8e-4)
    mean, _ = ops.bn_stats_from_partials(partial, parts, y.shape[0], Cout, None, None, 1e-5, 0.1)
    assert torch.allclose(mean.cpu(), cpu.mean(0), atol=3e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [
    (4, 64, 16, 16, 64, 3, 1, 1),          # rows 1024, cfg 256x128x32, M = 576 (ragged third tile), N = 64
    (2, 64, 16, 16, 128, 3, 2, 1),         # strided
    (4, 128, 8, 8, 256, 3, 1, 1),          # cfg 256x256x64
    (4, 64, 16, 16, 128, 1, 2, 0),         # 1x1 stride-2 downsample
    (2, 24, 8, 8, 72, 3, 1, 1),            # C % 8 only (tap boundaries inside a tile)
    (3, 256, 7, 7, 512, 3, 1, 1),          # 147 pixels: not a multiple of the k-tile (ragged last k-tile: zeros past the end)
    (7, 64, 6, 6, 64, 3, 1, 1),            # 252 pixels, 64-deep k-tiles
    (5, 128, 9, 5, 128, (3, 1), (2, 1), (1, 0)),   # temporal (3,1,1) / 2 of R(2+1)D over a [T, H*W] view: 125 output rows
    # products of more than 2^20 entries: the stand-alone scatter reduce goes through LDS-transposed tiles of 32 ci x taps x
    # 32 co (gemm.hip: splitk_reduce_conv_tiled_kernel) -- ragged tiles on both sides, and the 3-tap temporal form
    (2, 264, 6, 6, 520, 3, 1, 1),
    (2, 576, 4, 9, 640, (3, 1), 1, (1, 0)),
])
def test_implicit_gemm_weight_gradient_matches_explicit_path(dvt, device, dtype, geom):
    ops = dvt.ops
    N, Cin, H, W, Cout, k, stride, pad = geom
    g = torch.Generator().manual_seed(321)
    (kh, kw) = ops._pair(k)
    Ho, Wo = ops.conv_out_hw(H, W, k, stride, pad)
    # both operands sit in front of NaN-filled memory: a gather that reads past the last pixel row (the ragged k-tile of a
    # pixel count that is not a multiple of the k-tile) would poison the result
    xb = torch.full((N * H * W + 96, Cin), float("nan"), dtype=dtype, device="cuda")
    zb = torch.full((N * Ho * Wo + 96, Cout), float("nan"), dtype=dtype, device="cuda")
    x, dz = xb[: N * H * W], zb[: N * Ho * Wo]
    x.copy_(torch.randn(N * H * W, Cin, generator=g).to(dtype))
    dz.copy_(torch.randn(N * Ho * Wo, Cout, generator=g).to(dtype))
    assert ops.conv2d_implicit_wgrad_supported(x, dz, N, Cin, H, W, Cout, k, stride, pad)
    dwt = ops.conv2d_implicit_wgrad(x, dz, N, Cin, H, W, Cout, k, stride, pad)
    assert torch.isfinite(dwt).all()
    assert torch.equal(dwt, ops.conv2d_implicit_wgrad(x, dz, N, Cin, H, W, Cout, k, stride, pad))   # reproducible
    K = kh * kw * Cin
    col = ops.im2col(x, False, N, Cin, H, W, k, stride, pad, K, dtype)
    ref = col.float().t() @ dz.float()                         # [K, Cout]
    assert dwt.shape == ref.shape and rel_l2(dwt, ref) < 1e-5
    dw = ops.conv_weight_unpack_grad_t(dwt, (Cout, Cin, kh, kw))
    ref4 = ops.conv_weight_unpack_grad((dz.float().t() @ col.float()).contiguous(), (Cout, Cin, kh, kw))
    assert rel_l2(dw, ref4) < 1e-5
    acc = torch.ones_like(dw)
    ops.conv_weight_unpack_grad_t(dwt, (Cout, Cin, kh, kw), out=acc, accumulate=True)
    assert rel_l2(acc - 1.0, dw) < 1e-5
    # the split-K reduce scattering straight into the parameter's own layout: same sums, bit for bit -- stand-alone,
    # deferred + flushed, deferred + carried by the data-gradient launch, and accumulating into an existing gradient
    m1 = torch.full((Cout, Cin, kh, kw), float("nan"), device="cuda")
    ops.conv2d_implicit_wgrad(x, dz, N, Cin, H, W, Cout, k, stride, pad, master=m1)
    assert torch.equal(m1, dw)
    m2 = torch.full((Cout, Cin, kh, kw), float("nan"), device="cuda")
    _, pend = ops.conv2d_implicit_wgrad(x, dz, N, Cin, H, W, Cout, k, stride, pad, master=m2, defer_reduce=True)
    ops.splitk_reduce_pending(pend)
    assert torch.equal(m2, dw)
    m3 = torch.ones((Cout, Cin, kh, kw), device="cuda")
    _, pend = ops.conv2d_implicit_wgrad(x, dz, N, Cin, H, W, Cout, k, stride, pad, master=m3, defer_reduce=True, accumulate=True)
    if ops._pair(stride) == (1, 1) and Cin % 32 == 0:
        wd = ops.conv_weight_pack_dgrad(torch.randn(Cout, Cin, kh, kw, generator=g).cuda() * 0.05, dtype)
        (ph, pw) = ops._pair(pad)
        if ops.conv2d_implicit_supported(dz, wd, N, Cout, Ho, Wo, Cin, k, 1, (kh - 1 - ph, kw - 1 - pw)):
            ops.conv2d_implicit(dz, wd, N, Cout, Ho, Wo, Cin, k, 1, (kh - 1 - ph, kw - 1 - pw), carry=pend)
    ops.splitk_reduce_pending(pend)                      # (no-op when the data gradient carried it)
    assert rel_l2(m3 - 1.0, dw) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_grouped_weight_pack_is_the_two_transposes(dvt, device, dtype):
    """dvt_conv_weight_pack_group (once per optimizer step for every convolution of the encoder): kind 0 = forward operand
    [cout_p, ld] with column tap * cin_p + ci, kind 1 = data-gradient operand [cin_p, taps * cout_p] with rotated taps; the
    zero extension of channel-padded layers (R(2+1)D mid planes 45 / 230 / 921) and the K padding folded in.  Exact (a
    rounding of the fp32 parameter) against the index arithmetic written out in torch; shapes cover several rows per
    workgroup, one row per workgroup, ragged 64 x 64 tiles, a 7 x 7 stem and more entries than one launch holds."""
    ops = dvt.ops
    g = torch.Generator().manual_seed(17)
    shapes = [(64, 64, 3, 3, 64, 64, 576), (45, 3, 7, 7, 48, 3, 192), (230, 128, 3, 3, 240, 128, 1152),
              (128, 230, 3, 1, 128, 240, 720), (512, 512, 3, 3, 512, 512, 4608), (921, 512, 1, 1, 928, 512, 512),
              (70, 33, 2, 3, 80, 40, 256), (8, 8, 1, 1, 8, 8, 64)] * 6        # 96 entries: three launches
    entries, want = [], []
    for i, (co, ci, kh, kw, cop, cip, ld) in enumerate(shapes):
        w = torch.randn(co, ci, kh * kw, generator=g).cuda()
        taps = kh * kw
        wz = torch.zeros(cop, cip, taps, device="cuda")
        wz[:co, :ci] = w
        kind = (i // 8) % 2 if i >= 8 else i % 2
        for kd in ((0, 1) if i < 8 else (kind,)):
            if kd == 0:
                dst = torch.full((cop, ld), float("nan"), dtype=dtype, device="cuda")
                ref = torch.zeros(cop, ld, device="cuda")
                ref[:, :taps * cip] = wz.permute(0, 2, 1).reshape(cop, taps * cip)
            else:
                dst = torch.full((cip, taps * cop), float("nan"), dtype=dtype, device="cuda")
                ref = wz.flip(2).permute(1, 2, 0).reshape(cip, taps * cop)
            entries.append((w, dst, co, ci, kh, kw, cop, cip, ld, kd))
            want.append(ref.to(dtype))
    ops.conv_weight_pack_group(entries)
    for e, ref in zip(entries, want):
        assert torch.equal(e[1], ref), e[2:]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W", [(3, 56, 56), (2, 28, 28), (2, 13, 20), (1, 5, 56), (300, 8, 8), (70, 56, 56)])
def test_conv3x3_c64_weight_gradient_from_lds_halo_patches(dvt, device, dtype, N, H, W):
    """dvt_conv3x3_c64_wgrad (layer 1 of ResNet-18: 64 -> 64, 3 x 3 / 1 / 1): input patch and gradient tile staged once per
    tile, taps read as shifted windows of the patch, one fp32 partial per workgroup of the persistent grid, summed into the
    parameter layout by the split-K reduce -- against the implicit weight gradient (same products, other summation order)
    and against torch's conv2d weight gradient in fp32 on the same rounded operands; ragged last tiles, maps narrower and
    wider than a tile, more tiles than workgroups, accumulate, deferred reduce."""
    ops = dvt.ops
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn(N * H * W, 64, generator=g).to(dtype).cuda()
    dz = (torch.randn(N * H * W, 64, generator=g) / 8).to(dtype).cuda()
    assert ops.conv3x3_c64_wgrad_supported(x, dz, N, H, W)
    dw = torch.full((64, 64, 3, 3), float("nan"), device="cuda")
    ops.conv3x3_c64_wgrad(x, dz, N, H, W, dw)
    xr = x.float().view(N, H, W, 64).permute(0, 3, 1, 2)
    zr = dz.float().view(N, H, W, 64).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xr, (64, 64, 3, 3), zr, stride=1, padding=1)
    assert torch.isfinite(dw).all() and rel_l2(dw, ref) < 2e-5
    imp = torch.empty(64, 64, 3, 3, device="cuda")
    ops.conv2d_implicit_wgrad(x, dz, N, 64, H, W, 64, 3, 1, 1, master=imp)
    assert rel_l2(dw, imp) < 2e-5
    # accumulate into an existing gradient, through a deferred reduce resolved explicitly
    old = torch.randn(64, 64, 3, 3, generator=g).cuda()
    acc = old.clone()
    pend = ops.conv3x3_c64_wgrad(x, dz, N, H, W, acc, accumulate=True, defer_reduce=True)
    ops.splitk_reduce_pending(pend)
    assert rel_l2(acc - old, ref) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W,Cout", [(3, 56, 56, 144), (2, 13, 20, 144), (300, 8, 8, 144), (2, 28, 28, 80), (2, 28, 28, 128), (2, 28, 28, 96),
                                         (1, 14, 14, 160)])
def test_conv3x3_c64_weight_gradient_with_more_output_channels(dvt, device, dtype, N, H, W, Cout):
    """dvt_conv3x3_c64_wgrad_wide (the 64 -> 144 spatial half of R(2+1)D-18's layer-1 Conv2Plus1D): the halo-patch kernel once
    per channel group of a dz whose rows are Cout channels long (groups of 64, a last group of up to 80 on five blocks of 16 with
    the fifth block's own LDS image, narrower tails on the 64-wide kernel), every group summed
    into its rows of the parameter layout over the same workspace; against torch's fp32 conv2d weight gradient on the same
    operands; accumulate through a deferred reduce of the last group."""
    ops = dvt.ops
    g = torch.Generator().manual_seed(N * 100 + H + Cout)
    x = torch.randn(N * H * W, 64, generator=g).to(dtype).cuda()
    dz = (torch.randn(N * H * W, Cout, generator=g) / 8).to(dtype).cuda()
    assert ops.conv3x3_c64_wgrad_supported(x, dz, N, H, W, Cout)
    dw = torch.full((Cout, 64, 3, 3), float("nan"), device="cuda")
    ops.conv3x3_c64_wgrad(x, dz, N, H, W, dw, Cout=Cout)
    xr = x.float().view(N, H, W, 64).permute(0, 3, 1, 2)
    zr = dz.float().view(N, H, W, Cout).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xr, (Cout, 64, 3, 3), zr, stride=1, padding=1)
    assert torch.isfinite(dw).all() and rel_l2(dw, ref) < 2e-5
    old = torch.randn(Cout, 64, 3, 3, generator=g).cuda()
    acc = old.clone()
    pend = ops.conv3x3_c64_wgrad(x, dz, N, H, W, acc, accumulate=True, defer_reduce=True, Cout=Cout)
    ops.splitk_reduce_pending(pend)
    assert rel_l2(acc - old, ref) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,T,H,W", [(2, 12, 8, 7), (3, 12, 56, 56), (40, 12, 8, 8), (1, 8, 4, 4), (5, 16, 4, 6)])
def test_temporal_weight_gradient_from_lds_sliding_windows(dvt, device, dtype, N, T, H, W):
    """dvt_conv3x1_wgrad: weight gradient of the (3, 1, 1) temporal convolution 144 -> 64 of R(2+1)D-18's layer 1
    (torchvision r2plus1d_18, frame_transformer.py:64-74) with a segment of pixels over all frames staged once in LDS and the
    three taps read at three position offsets, against fp32 on the same 16-bit operands and against the implicit kernel it
    replaces; more tiles than workgroups, a map of a single tile column, accumulate, deferred reduce."""
    ops = dvt.ops
    Lp = H * W
    g = torch.Generator().manual_seed(N * 100 + T + Lp)
    x = torch.randn(N * T * Lp, 144, generator=g).to(dtype).cuda()
    dz = (torch.randn(N * T * Lp, 64, generator=g) / (N * T * Lp) ** 0.5).to(dtype).cuda()
    assert ops.conv3x1_wgrad_supported(x, dz, N, T, Lp, 144, 64)
    dw = torch.full((64, 144, 3, 1), float("nan"), device="cuda")
    ops.conv3x1_wgrad(x, dz, N, T, Lp, dw)
    xf, zf = x.float().view(N, T, Lp, 144), dz.float().view(N, T, Lp, 64)
    ref = torch.zeros(64, 144, 3, device="cuda")
    for kt in range(3):                                   # dW[co][ci][kt] = sum dz[n, t, p, co] * x[n, t + kt - 1, p, ci]
        lo, hi = max(0, 1 - kt), min(T, T + 1 - kt)
        ref[:, :, kt] = torch.einsum("ntpo,ntpi->oi", zf[:, lo:hi], xf[:, lo + kt - 1:hi + kt - 1])
    assert torch.isfinite(dw).all() and rel_l2(dw.view(64, 144, 3), ref) < 2e-5
    imp = torch.empty(64, 144, 3, 1, device="cuda")
    ops.conv2d_implicit_wgrad(x, dz, N, 144, T, Lp, 64, (3, 1), 1, (1, 0), master=imp)
    assert rel_l2(dw, imp) < 2e-5
    old = torch.randn(64, 144, 3, 1, generator=g).cuda()
    acc = old.clone()
    pend = ops.conv3x1_wgrad(x, dz, N, T, Lp, acc, accumulate=True, defer_reduce=True)
    ops.splitk_reduce_pending(pend)
    assert rel_l2(acc.view(64, 144, 3) - old.view(64, 144, 3), ref) < 2e-5


def _window_position_blocks(T, Lp):
    """16-position blocks per wave of the window forward for a clip geometry (csrc/conv3x1_window.h: window_plan picks the
    longest segment S <= 16 with Lp % S == 0 and (T * S) % 32 == 0 whose two windows fit; blocks = T * S / 32)."""
    for S in range(16, 1, -1):
        if Lp % S or (T * S) % 32:
            continue
        xbytes = (((T + 2) * S * 320) + 1023) & ~1023
        if 2 * xbytes + 4096 > 160 * 1024 or (xbytes >> 10) > 72:
            continue
        return T * S // 32 if 2 * xbytes >= 32768 else 0        # (tf_plan: the statistics scratch overlays the two windows)
    return 0


# one to six position blocks per wave: with one or two, the 16x16x16 tail of a tap follows the tap's last 16x16x32 step on
# the SAME accumulator with at most one other MFMA in between (csrc/conv3x1_fwd.hip: mfma_shape_fence)
_WINDOW_FWD_GEOMS = [(2, 12, 8, 7), (3, 12, 56, 56), (40, 12, 8, 8), (1, 8, 4, 4), (5, 16, 4, 6), (2, 8, 7, 8), (2, 20, 2, 4),
                     (3, 2, 4, 4), (300, 2, 4, 8)]


def test_window_forward_geometries_cover_every_block_count():
    assert {_window_position_blocks(T, H * W) for _, T, H, W in _WINDOW_FWD_GEOMS} == {1, 2, 3, 4, 5, 6}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,T,H,W", _WINDOW_FWD_GEOMS)
def test_temporal_forward_from_lds_sliding_windows_with_virtual_batchnorm(dvt, device, dtype, N, T, H, W):
    """dvt_conv3x1_fwd: the (3, 1, 1) temporal convolution 144 -> 64 of R(2+1)D-18's layer 1 from a window of all frames of a
    pixel segment, weights in registers -- against conv2d in fp32 on the same operands, with the BatchNorm partial sums of the
    stored output; and x_affine (both entry points): handed the spatial convolution's OUTPUT z and the BatchNorm + ReLU between
    the two halves, forward and weight gradient give bit for bit what they give on the materialised activation
    dvt_bn_apply_fwd writes (same formula, same rounding, zero rows of frames -1 and T kept zero)."""
    ops = dvt.ops
    Lp = H * W
    g = torch.Generator().manual_seed(N * 31 + T + Lp)
    z = torch.randn(N * T * Lp, 144, generator=g).to(dtype).cuda()
    w = (torch.randn(64, 144, 3, 1, generator=g) * (2.0 / (144 * 3)) ** 0.5).cuda()
    wp = ops.conv_weight_pack(w, ops.conv2d_implicit_k(144, 64, (3, 1)), dtype)
    assert ops.conv3x1_fwd_supported(z, wp, N, T, Lp, 144, 64)
    y, partial, parts = ops.conv3x1_fwd(z, wp, N, T, Lp, want_stats=True)
    zr = z.float().view(N, T, Lp, 144).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(zr, w.to(dtype).float(), None, 1, (1, 0)).permute(0, 2, 3, 1).reshape(-1, 64)
    assert y.shape == ref.shape and rel_l2(y, ref) < (5e-3 if dtype == torch.bfloat16 else 8e-4)
    mean, invstd = ops.bn_stats_from_partials(partial, parts, y.shape[0], 64, None, None, 1e-5, 0.1)
    assert torch.allclose(mean, y.float().mean(0), atol=2e-3) and rel_l2(invstd, (y.float().var(0, unbiased=False) + 1e-5).rsqrt()) < 2e-3
    assert torch.equal(ops.conv3x1_fwd(z, wp, N, T, Lp), y)
    # the BatchNorm + ReLU in front, virtual: (mean, invstd, gamma, beta) of a 144-plane layer
    m0, is0 = (0.2 * torch.randn(144, generator=g)).cuda(), (1 + 0.3 * torch.rand(144, generator=g)).cuda()
    ga, be = (1 + 0.2 * torch.randn(144, generator=g)).cuda(), (0.3 * torch.randn(144, generator=g)).cuda()
    act = ops.bn_apply_fwd(z, m0, is0, ga, be, None, True)                       # the materialised activation
    aff = (m0, is0, ga, be, 0, True)
    assert torch.equal(ops.conv3x1_fwd(z, wp, N, T, Lp, affine=aff), ops.conv3x1_fwd(act, wp, N, T, Lp))
    dz = (torch.randn(N * T * Lp, 64, generator=g) / (N * T * Lp) ** 0.5).to(dtype).cuda()
    if ops.conv3x1_wgrad_supported(z, dz, N, T, Lp, 144, 64):
        a, b = torch.empty(64, 144, 3, 1, device="cuda"), torch.empty(64, 144, 3, 1, device="cuda")
        ops.conv3x1_wgrad(z, dz, N, T, Lp, a, affine=aff)
        ops.conv3x1_wgrad(act, dz, N, T, Lp, b)
        assert torch.equal(a, b)
    nr = ops.bn_apply_fwd(z, m0, is0, ga, be, None, False)                       # without the ReLU
    assert torch.equal(ops.conv3x1_fwd(z, wp, N, T, Lp, affine=(m0, is0, ga, be, 0, False)), ops.conv3x1_fwd(nr, wp, N, T, Lp))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,T,H,W", [(2, 12, 8, 8), (3, 12, 56, 56), (40, 12, 4, 8), (1, 8, 4, 4), (5, 16, 4, 6), (2, 4, 7, 8), (1, 12, 2, 4),
                                     (3, 20, 2, 4)])
def test_stem_temporal_convolution_from_lds_sliding_windows(dvt, device, dtype, N, T, H, W):
    """dvt_conv3x1_fwd with 64 input channels: the temporal half of R(2+1)D-18's stem (torchvision r2plus1d_18 as used by
    frame_transformer.py:64-74: Conv3d(45, 64, (3, 1, 1)), the 45 mid planes stored zero-padded to 64) from a window of all
    frames of a 16-pixel segment -- against conv2d in fp32 on the same operands and the implicit kernel it replaces, with the
    BatchNorm partial sums of the stored output; and on the data-gradient pack of the weights (what the layer's backward
    launches) against the implicit data gradient.  Segments of 16, 12 and 8 pixels, one to six position blocks per wave, more
    tiles than workgroups."""
    ops = dvt.ops
    Lp = H * W
    g = torch.Generator().manual_seed(N * 13 + T + Lp)
    x = torch.randn(N * T * Lp, 64, generator=g).to(dtype)
    x[:, 45:] = 0                                                      # the padded planes
    x = x.cuda()
    w = torch.zeros(64, 64, 3, 1)
    w[:, :45] = torch.randn(64, 45, 3, 1, generator=g) * (2.0 / (45 * 3)) ** 0.5
    w = w.cuda()
    wp = ops.conv_weight_pack(w, ops.conv2d_implicit_k(64, 64, (3, 1)), dtype)
    assert ops.conv3x1_fwd_supported(x, wp, N, T, Lp, 64, 64)
    y, partial, parts = ops.conv3x1_fwd(x, wp, N, T, Lp, want_stats=True)
    xr = x.float().view(N, T, Lp, 64).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xr, w.to(dtype).float(), None, 1, (1, 0)).permute(0, 2, 3, 1).reshape(-1, 64)
    tol = 5e-3 if dtype == torch.bfloat16 else 8e-4
    assert y.shape == ref.shape and torch.isfinite(y.float()).all() and rel_l2(y, ref) < tol
    mean, invstd = ops.bn_stats_from_partials(partial, parts, y.shape[0], 64, None, None, 1e-5, 0.1)
    assert torch.allclose(mean, y.float().mean(0), atol=2e-3) and rel_l2(invstd, (y.float().var(0, unbiased=False) + 1e-5).rsqrt()) < 2e-3
    assert torch.equal(ops.conv3x1_fwd(x, wp, N, T, Lp), y)
    imp = ops.conv2d_implicit(x, wp, N, 64, T, Lp, 64, (3, 1), 1, (1, 0))
    assert rel_l2(y, imp) < tol
    # the layer's data gradient: the same kernel on the rotated / transposed pack
    dz = torch.randn(N * T * Lp, 64, generator=g).to(dtype).cuda()
    wd = ops.conv_weight_pack_dgrad(w, dtype)
    assert ops.conv3x1_fwd_supported(dz, wd, N, T, Lp, 64, 64)
    dx = ops.conv3x1_fwd(dz, wd, N, T, Lp)
    dr = dz.float().view(N, T, Lp, 64).permute(0, 3, 1, 2).requires_grad_(False)
    xg = xr.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xg, w.to(dtype).float(), None, 1, (1, 0)).backward(dr)
    dref = xg.grad.permute(0, 2, 3, 1).reshape(-1, 64)
    assert rel_l2(dx, dref) < tol and float(dx[:, 45:].float().abs().max()) == 0.0
    assert rel_l2(dx, ops.conv2d_implicit(dz, wd, N, 64, T, Lp, 64, (3, 1), 1, (1, 0))) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("N,T,H,W", [(2, 12, 8, 7), (3, 12, 56, 56), (1, 8, 4, 4)])
def test_temporal_data_gradient_with_the_mid_batchnorm_backward_fused(dvt, device, dtype, training, N, T, H, W):
    """dvt_conv3x1_stream_bn_bwd: the data gradient of the (3, 1, 1) temporal convolution 144 -> 64 with the backward of the
    BatchNorm + ReLU in front of that layer (the mid-plane BatchNorm of Conv2Plus1D) in its epilogue -- the 144-plane gradient
    is computed twice (sums, then the corrected gradient) and never stored -- against the composition it replaces on the same
    operands: dvt_conv3x1_stream, then dvt_bn_bwd with the mask recomputed from z.  dgamma / dbeta sum the same rounded values
    in another order; dz differs by the association of the fp32 correction (A dz + B z + C0) before the one rounding."""
    ops = dvt.ops
    Lp = H * W
    g = torch.Generator().manual_seed(N * 17 + T + Lp)
    rows = N * T * Lp
    z = torch.randn(rows, 144, generator=g).to(dtype).cuda()
    dy = (torch.randn(rows, 64, generator=g) / rows ** 0.5).to(dtype).cuda()
    w = (torch.randn(64, 144, 3, 1, generator=g) * (2.0 / (144 * 3)) ** 0.5).cuda()
    wd = ops.conv_weight_pack_dgrad(w, dtype)
    if not ops.conv3x1_stream_supported(dy, wd, N, T, Lp, 64, 144):
        pytest.skip("no segment length of this map fills half a 224-pixel tile")
    m0, is0 = (0.2 * torch.randn(144, generator=g)).cuda(), (1 + 0.3 * torch.rand(144, generator=g)).cuda()
    ga, be = (1 + 0.2 * torch.randn(144, generator=g)).cuda(), (0.3 * torch.randn(144, generator=g)).cuda()
    d = ops.conv3x1_stream(dy, wd, N, T, Lp, 64, 144)
    want, _, wg, wb = ops.bn_bwd(d, z, None, m0, is0, ga, True, training, False, beta=be)
    got, gg, gb = ops.conv3x1_stream_bn_bwd(dy, wd, z, (m0, is0, ga, be, 0, True), N, T, Lp, training)
    assert rel_l2(gg, wg) < 2e-5 and rel_l2(gb, wb) < 2e-5
    assert rel_l2(got, want) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    # accumulate: += into the caller's dgamma / dbeta
    ag, ab = torch.ones(144, device="cuda"), torch.full((144,), 2.0, device="cuda")
    got2, _, _ = ops.conv3x1_stream_bn_bwd(dy, wd, z, (m0, is0, ga, be, 0, True), N, T, Lp, training, dgamma=ag, dbeta=ab,
                                           accumulate=True)
    assert torch.equal(got2, got) and rel_l2(ag - 1, gg) < 1e-5 and rel_l2(ab - 2, gb) < 1e-5
    # without the ReLU
    want3, _, wg3, _ = ops.bn_bwd(d, z, None, m0, is0, ga, False, training, False)
    got3, gg3, _ = ops.conv3x1_stream_bn_bwd(dy, wd, z, (m0, is0, ga, be, 0, False), N, T, Lp, training)
    assert rel_l2(gg3, wg3) < 2e-5 and rel_l2(got3, want3) < (4e-3 if dtype == torch.bfloat16 else 6e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("N,T,H,W", [(2, 12, 8, 7), (3, 12, 56, 56), (1, 8, 4, 4)])
def test_fused_mid_batchnorm_backward_against_fp32_autograd(dvt, device, dtype, training, N, T, H, W):
    """dvt_conv3x1_stream_bn_bwd against an INDEPENDENT fp32 chain on the same 16-bit operands: torch's
    batch_norm -> relu -> conv2d (the (3, 1, 1) temporal convolution of torchvision's Conv2Plus1D, frame_transformer.py:64-74,
    over the [T, H*W] view) differentiated by autograd -- dz, dgamma, dbeta.  Training mode: the statistics handed to the
    kernel are the batch statistics of z (what the forward's statistics kernels produce), eval mode: arbitrary running ones."""
    ops = dvt.ops
    Lp = H * W
    g = torch.Generator().manual_seed(N * 19 + T + Lp)
    rows = N * T * Lp
    z = torch.randn(rows, 144, generator=g).to(dtype).cuda()
    dy = (torch.randn(rows, 64, generator=g) / rows ** 0.5).to(dtype).cuda()
    w = (torch.randn(64, 144, 3, 1, generator=g) * (2.0 / (144 * 3)) ** 0.5).cuda()
    wd = ops.conv_weight_pack_dgrad(w, dtype)
    if not ops.conv3x1_stream_supported(dy, wd, N, T, Lp, 64, 144):
        pytest.skip("no segment length of this map fills half a 224-pixel tile")
    ga, be = (1 + 0.2 * torch.randn(144, generator=g)).cuda(), (0.3 * torch.randn(144, generator=g)).cuda()
    eps = 1e-5
    if training:
        z64 = z.double()
        mean = z64.mean(0)
        var = (z64 - mean).pow(2).mean(0)
        m0, is0 = mean.float(), (var + eps).rsqrt().float()
    else:
        m0, is0 = (0.2 * torch.randn(144, generator=g)).cuda(), (1 + 0.3 * torch.rand(144, generator=g)).cuda()
    got, gg, gb = ops.conv3x1_stream_bn_bwd(dy, wd, z, (m0, is0, ga, be, 0, True), N, T, Lp, training)
    # the fp32 chain: [N, 144, T, Lp] maps, the temporal convolution as a (3, 1) conv2d with padding (1, 0)
    zf = z.float().view(N, T, Lp, 144).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    gaf, bef = ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
    if training:
        bn = TF.batch_norm(zf, None, None, gaf, bef, True, 0.1, eps)
    else:
        bn = TF.batch_norm(zf, m0.clone(), (1.0 / is0.double() ** 2 - eps).float(), gaf, bef, False, 0.1, eps)
    out = TF.conv2d(torch.relu(bn), w.to(dtype).float(), None, 1, (1, 0))
    out.backward(dy.float().view(N, T, Lp, 64).permute(0, 3, 1, 2))
    want = zf.grad.permute(0, 2, 3, 1).reshape(rows, 144)
    tol = 4e-3 if dtype == torch.bfloat16 else 6e-4
    assert torch.isfinite(got.float()).all() and rel_l2(got, want) < tol
    # dgamma / dbeta: the kernel sums the 144-plane gradient ROUNDED to the map's 16-bit type (what the unfused composition
    # stores and bn_bwd then reads), the fp32 chain the unrounded one: the element type's tolerance, not fp32's
    # (measured 1.4e-3 - 1.9e-3 in bf16, 1.9e-4 - 2.3e-4 in fp16)
    assert rel_l2(gg, gaf.grad) < tol and rel_l2(gb, bef.grad) < tol


def test_r2plus1d_block_backward_with_and_without_the_fused_mid_batchnorm(dvt, device, monkeypatch):
    """models/video_resnet.BasicBlock (layer 1 of R(2+1)D-18, frame_transformer.py:64-74): with the virtual BatchNorm between the
    halves of each Conv2Plus1D, the backward of that BatchNorm runs inside the temporal half's data gradient
    (functional.FUSED_MID_BN_BWD) or as dvt_bn_bwd over the stored gradient -- same input gradient, same parameter gradients
    (the mid BatchNorm's dgamma / dbeta included)."""
    from dvt_amd.models import video_resnet as vr
    F = dvt.functional
    torch.manual_seed(11)
    blk = vr.BasicBlock(64, 64, vr.Conv2Plus1D).cuda().train()
    N, T, H, W = 2, 12, 8, 7
    x0 = torch.randn(N * T * H * W, 64).to(torch.bfloat16).cuda()
    gy = (torch.randn(N * T * H * W, 64) / 50).to(torch.bfloat16).cuda()
    assert vr._virtual_bn_pair(blk.conv1[0], N, T, H, W, torch.bfloat16)
    res = []
    for fused in (False, True):
        monkeypatch.setattr(F, "FUSED_MID_BN_BWD", fused)
        for p_ in blk.parameters():
            p_.grad = None
        x = x0.clone().requires_grad_(True)
        out = blk.forward_ndhwc((x, N, T, H, W), torch.bfloat16)[0]
        out.backward(gy)
        res.append((x.grad.float().clone(), {n: p_.grad.clone() for n, p_ in blk.named_parameters()}))
    (dx0, g0), (dx1, g1) = res
    assert rel_l2(dx1, dx0) < 1e-2
    assert set(g0) == set(g1) and len(g0) == 12
    for n in g0:
        assert rel_l2(g1[n], g0[n]) < (1e-2 if n.endswith("weight") and g0[n].dim() > 1 else 2e-2), n
    # a switch flipped BETWEEN forward and backward (the forward has kept the mid BatchNorm virtual; the window and halo
    # kernels are then refused in backward): the backward materialises relu(bn(z)) and takes the implicit kernels -- same
    # gradients, no error
    for p_ in blk.parameters():
        p_.grad = None
    x = x0.clone().requires_grad_(True)
    out = blk.forward_ndhwc((x, N, T, H, W), torch.bfloat16)[0]
    monkeypatch.setattr(F, "HALO_CONV", False)
    out.backward(gy)
    assert rel_l2(x.grad, dx0) < 1e-2
    for n, p_ in blk.named_parameters():
        assert rel_l2(p_.grad, g0[n]) < (1e-2 if n.endswith("weight") and g0[n].dim() > 1 else 2e-2), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,W", [(16, 16), (9, 11)])
def test_col2im_joins_a_compact_downsample_gradient(dvt, device, dtype, H, W):
    """The first block of a ResNet stage (custom_resnet.py:121-136): its input feeds a 3x3 / 2 convolution and a 1x1 / 2
    downsample convolution.  dvt_col2im sums the 3x3 path's adjoint gather and the downsample's COMPACT input gradient
    (added at the pixels the stride-2 subsampling reads) in one pass == col2im + zero-filled col2im + add."""
    ops = dvt.ops
    g = torch.Generator().manual_seed(5)
    N, C = 3, 16
    Ho, Wo = ops.conv_out_hw(H, W, 3, 2, 1)
    dcol = torch.randn(N * Ho * Wo, 9 * C, generator=g).to(dtype).cuda()
    Hs, Ws = (H + 1) // 2, (W + 1) // 2
    dsub = torch.randn(N * Hs * Ws, C, generator=g).to(dtype).cuda()
    got = ops.col2im(dcol, N, C, H, W, 3, 2, 1, add=dsub, add_stride=2)
    a, b = ops.col2im(dcol, N, C, H, W, 3, 2, 1), ops.col2im(dsub, N, C, H, W, 1, 2, 0)
    want = a.float() + b.float()
    assert float((got.float() - want).abs().max()) <= (1e-6 if dtype == torch.float32 else 8e-3) * float(want.abs().max())
    full = torch.randn(N * H * W, C, generator=g).to(dtype).cuda()
    got2 = ops.col2im(dcol, N, C, H, W, 3, 2, 1, add=full)
    want2 = a.float() + full.float()
    assert float((got2.float() - want2).abs().max()) <= (1e-6 if dtype == torch.float32 else 8e-3) * float(want2.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [
    # N, Cin, H, W, Cout, k, stride, pad, shortcut ("alias": full-size second gradient; 2: compact gradient of a 1x1 / 2 shortcut)
    (3, 64, 16, 16, 128, 3, 2, 1, None), (2, 64, 9, 11, 128, 3, 2, 1, "alias"), (2, 128, 14, 14, 256, 3, 2, 1, 2),
    (5, 256, 7, 7, 512, 3, 2, 1, 2), (2, 64, 12, 49, 64, (3, 1), (2, 1), (1, 0), None), (3, 288, 7, 16, 128, (3, 1), (2, 1), (1, 0), "alias"),
    (40, 64, 28, 28, 128, 3, 2, 1, 2)])
def test_strided_data_gradient_by_parity_classes(dvt, device, dtype, geom):
    """dvt_conv_desc.out_h / out_w / out_rows + dvt_pack_entry kind 2: the data gradient of a STRIDED convolution
    (custom_resnet.py:19-22 with stride 2 and the 1x1 / 2 shortcut of :124-130; R(2+1)D's (3, 1) / (2, 1) temporal halves) as
    one implicit launch per parity class of input pixels, scattered into the full-size gradient with the shortcut's gradient
    (full-size, or the compact one of a strided 1x1) joining in the residual epilogue == conv_transpose in fp64 on the same
    16-bit operands, and == the dcol GEMM + col2im path it replaces (same operands, fp32 accumulation on both sides)."""
    ops = dvt.ops
    N, Cin, H, W, Cout, k, stride, pad, short = geom
    (kh, kw), (sh, sw), (ph, pw) = ops._pair(k), ops._pair(stride), ops._pair(pad)
    g = torch.Generator().manual_seed(N * 13 + Cin + H)
    Ho, Wo = ops.conv_out_hw(H, W, k, stride, pad)
    w = (torch.randn(Cout, Cin, kh, kw, generator=g) / (Cout * kh * kw / (sh * sw)) ** 0.5).cuda()
    dz = torch.randn(N * Ho * Wo, Cout, generator=g).to(dtype).cuda()
    full = compact = None
    if short == "alias":
        full = torch.randn(N * H * W, Cin, generator=g).to(dtype).cuda()
    elif short:
        Hs, Ws = (H - 1) // short + 1, (W - 1) // short + 1
        compact = torch.randn(N * Hs * Ws, Cin, generator=g).to(dtype).cuda()
    classes = ops.strided_dgrad_classes(k, stride, pad, H, W)
    assert classes is not None and len(classes) == sh * sw
    dx = torch.full((N * H * W, Cin), float("nan"), dtype=dtype, device="cuda")      # every row must be written by some class
    entries, wcs = [], []
    for (a, b, nt, pq, (rh, rw), hq) in classes:
        wc = torch.empty((Cin, nt[0] * nt[1] * Cout), dtype=dtype, device="cuda")
        entries.append((w.contiguous(), wc, Cout, Cin, kh, kw, Cout, Cin, 0, 2, (sh, sw, rh, rw)))
        wcs.append(wc)
    ops.conv_weight_pack_group(entries)
    for (a, b, nt, pq, _, hq), wc in zip(classes, wcs):
        res, rc = (full, False) if full is not None else ((compact, True) if (compact is not None and (a, b) == (0, 0)) else (None, False))
        ops.conv2d_implicit(dz, wc, N, Cout, Ho, Wo, Cin, nt, 1, pq, out=dx, out_hw=hq,
                            out_rows=ops.strided_class_rows(N, H, W, sh, sw, a, b, dz.device), residual=res, residual_compact=rc)
    assert torch.isfinite(dx.float()).all()
    # fp64 adjoint on the same rounded operands
    w64 = w.to(dtype).double().cpu()
    dz64 = dz.double().cpu().view(N, Ho, Wo, Cout).permute(0, 3, 1, 2)
    x0 = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    TF.conv2d(x0, w64, None, (sh, sw), (ph, pw)).backward(dz64)
    want = x0.grad.permute(0, 2, 3, 1).reshape(-1, Cin).clone()
    if full is not None:
        want += full.double().cpu()
    if compact is not None:
        wv = want.view(N, H, W, Cin)
        wv[:, ::short, ::short] += compact.double().cpu().view(N, Hs, Ws, Cin)
    tol = 4e-3 if dtype == torch.bfloat16 else 6e-4          # the output's own rounding
    assert rel_l2(dx, want) < tol
    # the path it replaces: dcol GEMM + col2im (+ the joined shortcut gradient)
    wp = ops.conv_weight_pack(w, kh * kw * Cin, dtype)
    dcol = ops.linear_dgrad(dz, wp)
    if full is not None:
        old = ops.col2im(dcol, N, Cin, H, W, k, stride, pad, add=full)
    elif compact is not None:
        old = ops.col2im(dcol, N, Cin, H, W, k, stride, pad, add=compact, add_stride=int(short))
    else:
        old = ops.col2im(dcol, N, Cin, H, W, k, stride, pad)
    assert rel_l2(dx, old) < 2 * tol                          # (the old path rounds dcol to 16 bits before it sums the taps)


@pytest.mark.parametrize("short", ["none", "alias", "compact"])
def test_conv_block_backward_with_and_without_strided_class_launches(dvt, device, monkeypatch, short):
    """functional._ConvBnAct.backward routes the data gradient of a strided layer through the parity-class launches
    (STRIDED_IMPLICIT, maps of at least STRIDED_IMPLICIT_MIN_PIXELS) or through dcol GEMM + col2im: same input gradient (the
    shortcut's gradient joined: none / the full-size one of fork="alias" / the compact one of a strided 1x1 shortcut), same
    parameter gradients -- the weight gradient's split-K reduce is carried by the last class launch."""
    F = dvt.functional
    N, Cin, H, W, Cout = 4, 64, 18, 14, 128
    g = torch.Generator().manual_seed(41)
    conv = torch.nn.Conv2d(Cin, Cout, 3, 2, 1, bias=False).cuda()
    bn = torch.nn.BatchNorm2d(Cout).cuda().train()
    with torch.no_grad():
        conv.weight.copy_((torch.randn(conv.weight.shape, generator=g) * (2.0 / (Cin * 9)) ** 0.5).cuda())
    x0 = torch.randn(N * H * W, Cin, generator=g).to(torch.bfloat16).cuda()
    Ho, Wo = dvt.ops.conv_out_hw(H, W, 3, 2, 1)
    gy = torch.randn(N * Ho * Wo, Cout, generator=g).to(torch.bfloat16).cuda()
    gs = torch.randn(N * (H * W if short == "alias" else Ho * Wo), Cin, generator=g).to(torch.bfloat16).cuda()
    res = []
    for floor in (1 << 40, 0):                       # class launches off (nothing is large enough) / on for every size
        monkeypatch.setattr(F, "STRIDED_IMPLICIT_MIN_PIXELS", floor)
        x = x0.clone().requires_grad_(True)
        conv.weight.grad = bn.weight.grad = bn.bias.grad = None
        fork = {"none": None, "alias": "alias", "compact": 2}[short]
        out = F.conv_bn_act(x, conv, bn, (N, Cin, H, W, False), relu=True, dtype=torch.bfloat16, fork=fork)
        if fork is None:
            out.backward(gy)
        else:
            y, second = out
            torch.autograd.backward([y, second], [gy, gs])
        res.append((x.grad.float().clone(), conv.weight.grad.clone(), bn.weight.grad.clone()))
    (dx0, dw0, dg0), (dx1, dw1, dg1) = res
    assert rel_l2(dx1, dx0) < 8e-3                   # (the explicit path rounds dcol to bf16 before it sums the taps)
    assert torch.equal(dw1, dw0) and torch.equal(dg1, dg0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_channel_padding_is_exact_zero_extension(dvt, device, dtype, tol):
    """cpad: a 24 -> 45 convolution followed by a 45 -> 32 one, run at padded widths 64 (R(2+1)D mid planes): the
    45 real channels, the running statistics and every parameter gradient equal the unpadded torch computation;
    the padded channels are exactly zero."""
    import torch.nn as nn
    F = dvt.functional
    torch.manual_seed(3)
    c1, b1 = nn.Conv2d(24, 45, 3, 1, 1, bias=False), nn.BatchNorm2d(45)
    c2, b2 = nn.Conv2d(45, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32)
    with torch.no_grad():
        for bn in (b1, b2):
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.2, 0.2)
    N, H, W = 4, 8, 8
    x = torch.randn(N, 24, H, W)
    ref_mods = [m for m in (c1, b1, c2, b2)]
    import copy
    r1, rb1, r2, rb2 = (copy.deepcopy(m) for m in ref_mods)
    yr = torch.relu(rb2(r2(torch.relu(rb1(r1(x))))))
    gy = torch.randn_like(yr)
    yr.backward(gy)
    for m in ref_mods:
        m.cuda()
    xm = x.permute(0, 2, 3, 1).reshape(-1, 24).to(dtype).cuda()
    h = F.conv_bn_act_raw(xm, c1.weight, b1, (N, 24, H, W, False), 3, 1, 1, relu=True, dtype=dtype, cpad=64)
    assert h.shape == (N * H * W, 64) and float(h.detach()[:, 45:].abs().max()) == 0.0
    y = F.conv_bn_act_raw(h, c2.weight, b2, (N, 64, H, W, False), 3, 1, 1, relu=True, dtype=dtype, cpad=32)
    assert y.shape == (N * H * W, 32)
    ref = yr.permute(0, 2, 3, 1).reshape(-1, 32)
    assert rel_l2(y, ref) < tol
    y.backward(gy.permute(0, 2, 3, 1).reshape(-1, 32).to(dtype).cuda())
    assert rel_l2(b1.running_mean, rb1.running_mean) < 5 * tol and rel_l2(b1.running_var, rb1.running_var) < 5 * tol
    for got, want in ((c1.weight, r1.weight), (b1.weight, rb1.weight), (b1.bias, rb1.bias), (c2.weight, r2.weight),
                      (b2.bias, rb2.bias)):
        assert got.grad.shape == want.grad.shape and rel_l2(got.grad, want.grad) < 10 * tol


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W,Cout,k,stride,pad", [(6, 64, 28, 28, 64, 3, 1, 1), (3, 64, 30, 26, 128, 3, 2, 1),
                                                        (40, 128, 14, 14, 256, 3, 1, 1), (5, 64, 56, 56, 64, (1, 3), 1, (0, 1)),
                                                        (8, 64, 56, 56, 144, 3, 1, 1)])
def test_conv_epilogue_batchnorm_statistics(dvt, device, N, C, H, W, Cout, k, stride, pad):
    """Column sums / sums of squares left by the implicit-convolution epilogue (incl. the > 256-part fold) against the
    stand-alone statistics pass over the stored output: mean, invstd and the running-statistics update."""
    ops = dvt.ops
    g = torch.Generator().manual_seed(3)
    kh, kw = ops._pair(k)
    x = torch.randn(N * H * W, C, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(Cout, C, kh, kw, generator=g) / (C * kh * kw) ** 0.5).cuda()
    wp = ops.conv_weight_pack(w, kh * kw * C, torch.bfloat16)
    if not ops.conv2d_implicit_supported(x, wp, N, C, H, W, Cout, k, stride, pad):
        pytest.skip("geometry not served by the implicit kernel")
    z, partial, parts = ops.conv2d_implicit(x, wp, N, C, H, W, Cout, k, stride, pad, want_stats=True)
    z_plain = ops.conv2d_implicit(x, wp, N, C, H, W, Cout, k, stride, pad)
    assert torch.equal(z, z_plain)
    rm1, rv1 = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    rm2, rv2 = rm1.clone(), rv1.clone()
    mean, invstd = ops.bn_stats_from_partials(partial, parts, z.shape[0], Cout, rm1, rv1, 1e-5, 0.1)
    mean_ref, invstd_ref = ops.bn_stats(z, rm2, rv2, 1e-5, 0.1)
    zf = z.float()
    # the epilogue sums the fp32 accumulators, the stand-alone pass the bf16-rounded output: equal to rounding noise
    assert float((mean - mean_ref).abs().max()) < 2e-3 * float(zf.std()) + 1e-6
    assert float((invstd / invstd_ref - 1).abs().max()) < 2e-3
    assert float((rm1 - rm2).abs().max()) < 1e-3 and float((rv1 / rv2 - 1).abs().max()) < 1e-3


def test_packed_weights_follow_edits_behind_the_stores_back(dvt, device, monkeypatch):
    """dp.FlatParameters.packed_weight: the cached GEMM-operand forms of the convolution weights are refreshed by
    ONE grouped launch after an optimizer step, and an edit the store did not make (load_state_dict, an in-place op under
    no_grad) repacks the forms of THAT parameter only -- one single-entry launch each, not a full-group launch per stale key --
    with outputs equal to a freshly built model holding the same weights."""
    from dvt_amd.dp import FlatParameters
    from dvt_amd.models.custom_resnet import resnet18
    ops = dvt.ops
    torch.manual_seed(5)
    net = resnet18(False, compute_dtype=torch.bfloat16).cuda().train()
    flat = FlatParameters(net, compute_dtype=torch.bfloat16)
    x = torch.randn(2, 3, 64, 64).cuda()
    calls = []
    real = ops.conv_weight_pack_group
    monkeypatch.setattr(ops, "conv_weight_pack_group", lambda entries: (calls.append(len(entries)), real(entries))[1])
    net(x)                                                   # first use: every form is created (one single-entry launch each)
    nforms = len(flat._packed)
    assert nforms >= 15 and calls == [1] * nforms
    calls.clear()
    net(x)
    assert calls in ([], [nforms])                           # (the cache starts out invalid: at most one grouped refresh)
    calls.clear()
    net(x)
    assert calls == []                                       # nothing changed: no launch
    flat.invalidate_packed()                                 # what an optimizer step does
    net(x)
    assert calls == [nforms]                                 # one grouped launch
    calls.clear()
    other = resnet18(False, compute_dtype=torch.bfloat16).cuda().train()
    with torch.no_grad():
        for p_ in other.parameters():
            p_.mul_(1.5)
    net.load_state_dict(other.state_dict())                  # bumps every parameter's version behind the store's back
    want = other(x)                                          # (a model outside a store packs on the spot: not counted)
    calls.clear()
    got = net(x)
    assert all(c == 1 for c in calls) and len(calls) == nforms, calls      # K single-entry launches, not K x K work
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    calls.clear()
    net(x)
    assert calls == []


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W,Cout", [(3, 224, 224, 64), (2, 112, 112, 45), (5, 32, 32, 64), (2, 36, 20, 64), (300, 16, 16, 64),
                                        (1, 64, 48, 64)])
def test_stem_convolution_from_an_lds_halo_patch(dvt, device, dtype, N, H, W, Cout):
    """dvt_conv_stem7: the 7x7 / 2 / 3 stem on 3-channel frames (custom_resnet.py:100 `conv1`; with 45 planes zero-extended to
    64 the (1, 7, 7) spatial half of the R(2+1)D stem behind frame_transformer.py:64-74) from an LDS halo patch of the
    pixel-pair map with the weights in registers -- against conv2d in fp32 on the same 16-bit operands, with the BatchNorm
    partial sums of the stored output, and against the implicit GEMM it replaces (same pair geometry: (7, 4) / (2, 1) /
    (3, 2), trim 1).  Whole frames per tile, ragged last tiles, more tiles than workgroups, rows narrower than a block."""
    ops = dvt.ops
    g = torch.Generator().manual_seed(N * 7 + H + W + Cout)
    x = torch.randn(N, 3, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(Cout, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5).cuda()
    xp = ops.nchw_to_nhwc_pad(x, dtype, 4)                                  # [N*H*(W/2), 8]
    w4 = w.reshape(Cout, 3, 49)
    if Cout != 64:
        w4 = ops.pad3_f32(w4, Cout, 3, 49, 64, 3)
    wpairs = ops.conv_weight_pairs(w4, 64, 3, 7, 7, 3, 4)                   # [64, 8, 7, 4]
    wp = ops.conv_weight_pack(wpairs, ops.conv2d_implicit_k(8, 64, (7, 4)), dtype)
    Wp, Ho = W // 2, H // 2
    assert ops.conv_stem7_supported(xp, wp, N, H, Wp)
    z, partial, parts = ops.conv_stem7(xp, wp, N, H, Wp, want_stats=True)
    ref = TF.conv2d(x.float(), w.to(dtype).float(), None, 2, 3).permute(0, 2, 3, 1).reshape(-1, Cout)
    tol = 5e-3 if dtype == torch.bfloat16 else 8e-4
    assert z.shape == (N * Ho * Wp, 64) and torch.isfinite(z.float()).all()
    assert rel_l2(z[:, :Cout], ref) < tol and float(z[:, Cout:].float().abs().max() if Cout < 64 else 0.0) == 0.0
    mean, invstd = ops.bn_stats_from_partials(partial, parts, z.shape[0], 64, None, None, 1e-5, 0.1)
    assert torch.allclose(mean, z.float().mean(0), atol=2e-3)
    assert rel_l2(invstd, (z.float().var(0, unbiased=False) + 1e-5).rsqrt()) < 2e-3
    assert torch.equal(ops.conv_stem7(xp, wp, N, H, Wp), z)
    imp = ops.conv2d_implicit(xp, wp, N, 8, H, Wp, 64, (7, 4), (2, 1), (3, 2), trim_w=1)
    assert imp.shape == z.shape and rel_l2(z, imp) < tol


@pytest.mark.parametrize("N,T,H,W", [(28, 12, 56, 56), (300, 12, 8, 8), (7, 8, 28, 28)])
def test_helper_wave_kernels_are_deterministic(dvt, device, N, T, H, W):
    """The sixteen-wave kernels of round 6 (compute waves + helper waves behind one barrier per tile: conv3x1_fwd_pipe,
    conv3x1_wgrad_pipe, conv3x1_dbn; torchvision's Conv2Plus1D behind frame_transformer.py:64-74) hand buffers between wave
    roles through counted vmcnt waits and barriers: a missing wait shows as run-to-run differences long before it shows in a
    tolerance.  Six launches of each on the same operands -- at the product shape, with many more tiles than workgroups, and
    with another frame count -- must agree bit for bit, with other kernels' traffic in between."""
    ops = dvt.ops
    Lp = H * W
    rows = N * T * Lp
    g = torch.Generator().manual_seed(5 + N + T)
    z = torch.randn(rows, 144, generator=g).to(torch.bfloat16).cuda()
    dy = (torch.randn(rows, 64, generator=g) / 8).to(torch.bfloat16).cuda()
    w = (torch.randn(64, 144, 3, 1, generator=g) * 0.05).cuda()
    wp = ops.conv_weight_pack(w, ops.conv2d_implicit_k(144, 64, (3, 1)), torch.bfloat16)
    wd = ops.conv_weight_pack_dgrad(w, torch.bfloat16)
    m0, is0 = (0.2 * torch.randn(144, generator=g)).cuda(), (1 + 0.3 * torch.rand(144, generator=g)).cuda()
    ga, be = (1 + 0.2 * torch.randn(144, generator=g)).cuda(), (0.3 * torch.randn(144, generator=g)).cuda()
    aff = (m0, is0, ga, be, 0, True)
    noise = torch.randn(1 << 22, device="cuda")
    first = None
    for rep in range(6):
        y, partial, parts = ops.conv3x1_fwd(z, wp, N, T, Lp, want_stats=True, affine=aff)
        st = partial[:parts * 2 * 64].clone()
        dw = torch.empty(64, 144, 3, 1, device="cuda")
        ops.conv3x1_wgrad(z, dy, N, T, Lp, dw, affine=aff)
        outs = [y, st, dw]
        if ops.conv3x1_stream_supported(dy, wd, N, T, Lp, 64, 144):
            outs += list(ops.conv3x1_stream_bn_bwd(dy, wd, z, aff, N, T, Lp, True))
        noise.mul_(1.0001)                                       # (unrelated traffic between the repetitions)
        if first is None:
            first = [o.clone() for o in outs]
        else:
            for a, b in zip(outs, first):
                assert torch.equal(a, b), rep
