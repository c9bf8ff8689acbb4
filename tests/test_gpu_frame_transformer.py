"""Parity of the FrameTransformer / PTN token paths (SURVEY section 8 rows a8, a9, a10 and
a15/a16 token parts, a17-a19) on the GPU against the golden vectors and the CPU oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import clip_path as O
from tests.util import golden, rel_l2

pytestmark = pytest.mark.gpu

T = lambda a: torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_transformer_base_matches_torch_encoder_golden(device, dtype, tol):
    """TransformerBase == nn.TransformerEncoder (post-norm, ReLU, seq-first, eval)."""
    from dvt_amd.models.frame_transformer import TransformerBase
    g = golden("encoder_postnorm.npz")
    net = TransformerBase(64, 64, int(g["nhead"]), 96, 2, 0.5)
    net.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w:")})
    net = net.cuda().eval()
    x = T(g["x"]).to(dtype).cuda().requires_grad_(True)
    y = net(x)
    assert rel_l2(y, T(g["y"])) < tol
    y.backward(T(g["gy"]).to(dtype).cuda())
    assert rel_l2(x.grad, T(g["gx"])) < 2 * tol


def test_positional_encoding_matches_reference(device):
    from dvt_amd.models.frame_transformer import PositionalEncoding
    g = golden("posenc.npz")
    for d, L in ((896, 14), (2048, 14), (64, 9)):
        pe = PositionalEncoding(d, 0.5, max_len=L).cuda().eval()
        assert torch.allclose(pe.pe.cpu(), T(g[f"pe_{d}_{L}"]), atol=1e-6)   # host libm may differ in the last ulp
        out = pe(torch.ones(L, 2, d, device="cuda"))
        assert torch.allclose(out.cpu(), T(g[f"fwd_{d}_{L}"]), atol=1e-6)
    with pytest.raises(ValueError, match="positional table"):
        pe(torch.ones(L + 1, 2, d, device="cuda"))
    pe.train()                                   # p = 0.5: half the entries dropped, survivors doubled
    out = pe(torch.ones(L, 2, d, device="cuda"))
    ev = pe.eval()(torch.ones(L, 2, d, device="cuda"))
    kept = out != 0
    assert 0.4 < float(kept.float().mean()) < 0.6
    assert torch.allclose(out[kept], 2.0 * ev[kept], atol=1e-6)


from tests.encoders import PatchLinearEncoder, oracle_patch_linear_encoder as _oracle_encoder


def _make_ft(mode, dtype):
    from dvt_amd.models.frame_transformer import FrameTransformer
    torch.manual_seed(1130)
    d = 64
    cfg = dict(batch_size=2, seq_len=4, cls=1, model=mode, opt="adamW", learning_rate=5e-6, weight_decay=0.09,
               momentum=0.005, d_model=d, tokens=5, frame_len=2, clip_size=16, img_size=32, vid_nhead=2,
               vid_nhid=96, scene_nhead=2, scene_nhid=96, compute_dtype=dtype,
               vid_encoder=PatchLinearEncoder(3, 8, 32, d, dtype), img_encoder=PatchLinearEncoder(3, 8, 32, d, dtype))
    net = FrameTransformer(**cfg)
    # the reference's 3-layer head has 512/128 hidden units (frame_transformer.py:106)
    return net.cuda().eval()


def _oracle_params(net):
    return {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()
            if v.dtype.is_floating_point}


def _oracle_vid_cls_embedding(P, vid):
    """vid_step (frame_transformer.py:192-210): CLS chunk + chunks -> encoder -> pos-enc -> distil_transformer -> CLS."""
    B = vid.shape[0]
    cls = P["vid_cls"]                                                       # [1, T, 3, H, W]
    data = torch.cat((cls.unsqueeze(0).expand(B, *cls.shape), vid), dim=1)   # [B, 5, T, 3, H, W]
    data = data.reshape(-1, *data.shape[2:]).permute(0, 2, 1, 3, 4)
    emb = _oracle_encoder(data, P, "vid_model.", 8).reshape(B, 5, -1).permute(1, 0, 2)
    seq = emb + P["position_encoder.pe"][:5]
    return O.transformer_base(seq, P, "distil_transformer.", 4, 2)[0]


def _oracle_ft_vid(net, vid):
    P = _oracle_params(net)
    return O.mlp_head3(_oracle_vid_cls_embedding(P, vid), P), P


def _oracle_ft_forward(P, mode, img, vid):
    """The intended semantics of FrameTransformer.forward per mode (frame_transformer.py:136-190, img_step :212-244;
    the deviations from the non-executable text are the ones listed in the module docstring of the build)."""
    B = img.shape[0]
    icls = P["img_cls"]
    idata = torch.cat((icls.unsqueeze(0).expand(B, *icls.shape), img), dim=1).reshape(-1, 3, 32, 32)
    iemb = _oracle_encoder(idata, P, "img_model.", 8).reshape(B, 5, -1).permute(1, 0, 2)
    if mode in ("frame", "pre_modal", "sum_residual"):
        seq = O.transformer_base(iemb + P["position_encoder.pe"][:5], P, "scene_transformer.", 4, 2)
        if mode == "sum_residual":                      # as executed (:149-161): 2 * normalize(img_cls)
            n = seq[0] / seq[0].norm(dim=-1, keepdim=True).clamp_min(1e-12)
            return O.mlp_head3(n + n, P)
        return O.mlp_head3(seq[0], P)
    vcls = _oracle_vid_cls_embedding(P, vid)
    joint = torch.cat((iemb, vcls.unsqueeze(0)), dim=0)                      # 6 tokens: the injected video CLS (:225-226)
    seq = O.transformer_base(joint + P["position_encoder.pe"][:6], P, "scene_transformer.", 4, 2)
    img_cls, vid_tkn = seq[0], seq[-1]
    if mode == "sum":
        return O.mlp_head3(img_cls + vid_tkn, P)
    if mode == "post_sum":
        return O.mlp_head3(img_cls + vcls, P)
    return O.mlp_head3(img_cls, P), O.mlp_head3(vid_tkn, P)                  # distil


def _oracle_ft_loss(P, mode, img, vid, target):
    out = _oracle_ft_forward(P, mode, img, vid)
    if mode == "distil":                                                     # :250-252
        s, t = out
        return O.bce_with_logits(s, target) + O.cross_entropy_hard(s, t), out
    return O.bce_with_logits(out, target), out


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_frame_transformer_vid_mode_fwd_bwd(device, dtype, tol):
    net = _make_ft("vid", dtype)
    g = torch.Generator().manual_seed(3)
    vid = torch.randn(2, 4, 2, 3, 16, 16, generator=g)
    img = torch.randn(2, 4, 3, 32, 32, generator=g)
    target = (torch.rand(2, 19, generator=g) < 0.3).float()
    ref_logits, P = _oracle_ft_vid(net, vid)
    ref_loss = O.bce_with_logits(ref_logits, target)
    ref_loss.backward()
    loss = net.training_step((target.cuda(), img.cuda(), vid.cuda()), 0)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < (1e-5 if dtype == torch.float32 else 5e-3)
    for k in ("vid_cls", "img_mlp_head.0.weight", "img_mlp_head.4.bias", "vid_model.embed.weight",
              "distil_transformer.transformer.layers.0.self_attn.in_proj_weight",
              "distil_transformer.transformer.layers.3.linear2.weight",
              "distil_transformer.transformer.layers.1.norm1.weight"):
        got = dict(net.named_parameters())[k].grad
        assert rel_l2(got, P[k].grad) < 3 * tol, k
    # the reference's surface
    assert net.hparams.seq_len == 5 and isinstance(net.running_logits, list)
    net.validation_step((target.cuda(), img.cuda(), vid.cuda()), 0)
    assert len(net.running_logits) == 1 and net.running_logits[0].shape == (2, 19)
    probs = net.running_logits[0].float().cpu()                       # sigmoid probabilities, integer labels (:331-334)
    assert float(probs.min()) >= 0.0 and float(probs.max()) <= 1.0
    assert rel_l2(probs, torch.sigmoid(ref_logits.detach())) < 5 * tol and net.running_labels[0].dtype == torch.int32
    # torchmetrics-style accumulators of the reference surface (train_aprc / val_aprc / cos)
    from oracle import eval_metrics as EM
    ap = torch.stack(net.val_aprc.compute()).cpu().numpy()
    want = EM.average_precision(ref_logits.detach().numpy(), target.numpy())[2]
    assert ap.shape == (19,) and np.abs(ap - want).max() < (1e-6 if dtype == torch.float32 else 0.26)   # 2 samples: coarse ranks
    c = net.cos(torch.randn(3, 19).cuda(), torch.randn(3, 19).cuda())
    assert c.shape == (3,) and float(c.abs().max()) <= 1.0 + 1e-6
    from dvt_amd.metrics import TransformerEval
    scalars = TransformerEval().on_validation_epoch_end(None, net)     # the callback consumes exactly these accumulators
    assert "sklearn apr" in scalars and net.running_logits == []


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["sum", "distil", "post_sum", "frame", "sum_residual", "pre_modal"])
def test_frame_transformer_cross_modal_modes_fwd_bwd(device, mode, dtype):
    """every image / cross-modal mode, forward AND backward.  ``sum`` / ``distil`` / ``post_sum``
    inject the video CLS embedding as one more token of the image sequence, which then self-attends jointly
    (frame_transformer.py:225-226), so the loss gradient reaches the video branch through the scene encoder's
    attention: compared are the logits, the loss (BCE, + hard-label CE in ``distil``, :250-252) and the gradient of
    EVERY parameter -- ``vid_cls``, ``img_cls``, both encoders, ``distil_transformer.*``, ``scene_transformer.*``, the
    head -- against the CPU oracle.  fp32 kernels: 5e-4.  bf16 kernels: <= 2x the oracle's own bf16 deviation on the
    same inputs (the oracle composition re-run in bf16, see below: the protocol of SURVEY section 7)."""
    net = _make_ft(mode, dtype)
    g = torch.Generator().manual_seed(4)
    vid = torch.randn(2, 4, 2, 3, 16, 16, generator=g)
    img = torch.randn(2, 4, 3, 32, 32, generator=g)
    target = (torch.rand(2, 19, generator=g) < 0.3).float()
    P = _oracle_params(net)
    ref_loss, ref_out = _oracle_ft_loss(P, mode, img, vid, target)
    ref_loss.backward()
    out = net(img.cuda(), vid.cuda())
    loss = net.training_step((target.cuda(), img.cuda(), vid.cuda()), 0)
    loss.backward()
    named = dict(net.named_parameters())
    if dtype == torch.float32:
        tol_out, bound = 2e-4, {k: 5e-4 for k in P}
        assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-4
    else:
        # the oracle's own bf16 runs on the same inputs: autocast (fp32 stream, bf16 matmuls) and plain .bfloat16()
        # (16-bit storage throughout, which is what the HIP path does).  At this toy size (12 rows x 96 hidden units per
        # post-norm ReLU layer) both are dominated by the same lottery: a pre-activation within rounding of zero flips
        # its ReLU mask and moves that layer's linear1 gradient by ~3e-2 per flip (measured: the .bfloat16() oracle
        # deviates by 0.16-0.22 on such a layer, autocast by 0.01-0.05; the HIP path sits between the two).
        Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lp_loss, lp_out = _oracle_ft_loss(Q, mode, img, vid, target)
        lp_loss.float().backward()
        R = {k: v.detach().bfloat16().requires_grad_(True) for k, v in P.items()}
        pb_loss, pb_out = _oracle_ft_loss(R, mode, img.bfloat16(), vid.bfloat16(), target.bfloat16())
        pb_loss.backward()
        first = lambda o: (o[0] if isinstance(o, tuple) else o).detach().float()
        tol_out = 2 * max(rel_l2(first(lp_out), first(ref_out)), rel_l2(first(pb_out), first(ref_out))) + 2e-3
        yard = {k: max(rel_l2(Q[k].grad, P[k].grad), rel_l2(R[k].grad.float(), P[k].grad))
                for k in P if P[k].grad is not None and float(P[k].grad.abs().max()) > 0}
        med = float(np.median(list(yard.values())))
        bound = {k: 2 * max(v, med) + 5e-3 for k, v in yard.items()}
        assert abs(float(loss.detach()) - float(ref_loss.detach())) < 5e-3
    outs, refs = (out if isinstance(out, tuple) else (out,)), (ref_out if isinstance(ref_out, tuple) else (ref_out,))
    for a, b in zip(outs, refs):
        assert rel_l2(a, b) < tol_out, (mode, rel_l2(a, b), tol_out)
    checked, worst = 0, ("", 0.0)
    for k, p in P.items():
        if k not in named:
            continue                                     # buffers (positional table)
        got = named[k].grad
        if p.grad is None or float(p.grad.abs().max()) == 0.0:
            assert got is None or float(got.abs().max()) == 0.0, (mode, k, "gradient where the oracle has none")
            continue
        assert got is not None, (mode, k)
        e = rel_l2(got, p.grad)
        worst = max(worst, (k, e / bound[k]), key=lambda t: t[1])
        assert e < bound[k], (mode, k, e, bound[k])
        checked += 1
    print(f"[ft/{mode}/{dtype}] {checked} parameter gradients checked; worst error/bound {worst[1]:.2f} ({worst[0]})")
    expect = {"frame": ("img_cls", "scene_transformer", "img_model"), "pre_modal": ("img_cls", "scene_transformer"),
              "sum_residual": ("img_cls", "scene_transformer")}.get(mode, ("img_cls", "vid_cls", "scene_transformer",
                                                                            "distil_transformer", "vid_model", "img_model"))
    for pre in expect + ("img_mlp_head",):
        assert any(k.startswith(pre) and P[k].grad is not None and float(P[k].grad.abs().max()) > 0 for k in P), pre
    assert checked >= (20 if mode in ("frame", "pre_modal", "sum_residual") else 40)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_simple_transformer_ptn(device, dtype, tol):
    from dvt_amd.models.transformer import SimpleTransformer
    torch.manual_seed(7)
    net = SimpleTransformer(batch_size=3, seq_len=4, cls=1, dropout=0.5, input_dimension=64, nhead=2, nhid=96,
                            nlayers=2, model="ptn", learning_rate=1e-3, momentum=0.9, weight_decay=0.0,
                            compute_dtype=dtype).cuda().eval()
    g = torch.Generator().manual_seed(8)
    data = torch.randn(3, 4, 3, 64, generator=g)                           # [B, S, E=3 experts, D]
    label = (torch.rand(3, 15, generator=g) < 0.3).float()
    P = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()
         if v.dtype.is_floating_point}
    total = 0
    for i in range(3):
        seq = torch.cat((P["cls"], data[:, :, i].permute(1, 0, 2)), dim=0) + P["position_encoder.pe"][:5]
        seq = O.layernorm(seq, P["norm.weight"], P["norm.bias"])
        if i < 2:
            seq = O.transformer_base(seq, {k.replace(f"transformer_encoder{i}.", "transformer."): v
                                           for k, v in P.items()}, "", 2, 2)
        total = total + seq[0]
    ref = O.linear(O.layernorm(total, P["mlp_head.0.weight"], P["mlp_head.0.bias"]), P["mlp_head.1.weight"],
                   P["mlp_head.1.bias"])
    ref_loss = O.bce_with_logits(ref, label)
    ref_loss.backward()
    loss = net.training_step({"experts": data.cuda(), "label": label.cuda()}, 0)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < (1e-5 if dtype == torch.float32 else 5e-3)
    for k in ("cls", "norm.weight", "transformer_encoder0.layers.1.linear1.weight",
              "transformer_encoder1.layers.0.self_attn.in_proj_bias", "mlp_head.1.weight"):
        assert rel_l2(dict(net.named_parameters())[k].grad, P[k].grad) < 3 * tol, k
