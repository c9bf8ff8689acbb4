"""BASELINE configs[2] / configs[3]: the 3-scale pyramid front-end feeding the
space -> time transformer, the cross-modal (video queries, audio keys/values; Lq != Lk) attention block and
the distillation head, on the GPU against the CPU oracle composition (oracle/pyramid_path.py)."""
import numpy as np
import pytest
import torch

from oracle import clip_path as O
from oracle import pyramid_path as PP
from tests.util import (rel_l2, golden, fill_pyramid_from_numpy, pyramid_digest_inputs, grad_digest_errors,
                        reference_lowprec_yardstick, assert_within_reference_lowprec)

pytestmark = pytest.mark.gpu


def _randomise(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() == 1 and name.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "token" in name or "pos_embedding" in name:
                p.copy_(torch.randn(p.shape, generator=g))
            elif p.dim() == 4:
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p.shape[1]) ** 0.5)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 6e-2)])
def test_cross_attention_block(device, dtype, tol):
    """Lq = 9 video tokens attend to Lk = 32 audio tokens (dh = 64: the MFMA attention kernel in bf16)."""
    from dvt_amd.models.pyramid_vivit import CrossAttention
    torch.manual_seed(0)
    blk = CrossAttention(128, heads=2, dim_head=64)
    _randomise(blk, 3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 9, 128, generator=g).requires_grad_(True)
    c = torch.randn(3, 32, 128, generator=g).requires_grad_(True)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    ref = PP.cross_attention(x, c, P, "", 2)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    blk = blk.cuda()
    xd = x.detach().to(dtype).cuda().requires_grad_(True)
    cd = c.detach().to(dtype).cuda().requires_grad_(True)
    out = blk(xd, cd)
    out.backward(gy.to(dtype).cuda())
    assert rel_l2(out, ref) < tol
    assert rel_l2(xd.grad, x.grad) < 2 * tol and rel_l2(cd.grad, c.grad) < 2 * tol
    for k, p in blk.named_parameters():
        assert rel_l2(p.grad, P[k].grad) < 3 * tol, k


class _F64View:
    """The ``f64:`` entries of a pyramid digest presented like a digest of their own (``files`` / ``[]``)."""

    def __init__(self, npz):
        self.d = {k[4:]: npz[k] for k in npz.files if k.startswith("f64:")}
        self.files = list(self.d)

    def __getitem__(self, k):
        return self.d[k]


def _oracle_run(clip, audio, target, state, pnames, variant, depth, heads, mode, training_bn):
    """The oracle composition in fp32 (mode None), under torch.autocast (``amp_<prec>``) or with parameters and inputs
    cast to the 16-bit type (``pure_<prec>``): (student, teacher, loss, {name: gradient})."""
    cm = variant == "crossmodal"
    kind, prec = mode.split("_") if mode else (None, None)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, None: torch.float32}[prec]
    scale = 1024.0 if prec == "fp16" else 1.0
    cast = (lambda t: t.to(dt)) if kind == "pure" else (lambda t: t)
    P = {k: (cast(v) if v.dtype.is_floating_point else v).clone() for k, v in state.items()}
    for k in pnames:
        P[k].requires_grad_(True)

    def fwd():
        return PP.pyramid_vivit_forward(cast(clip), cast(audio) if cm else None, P, depth=depth, heads=heads,
                                        training_bn=training_bn, distill=cm)

    if kind == "amp":
        with torch.autocast("cpu", dtype=dt):
            out = fwd()
    else:
        out = fwd()
    if cm:
        student, teacher = out[0].float(), out[1].float()
        loss = O.bce_with_logits(student, target) + O.cross_entropy_hard(student, teacher)
    else:
        student, teacher = out.float(), None
        loss = O.bce_with_logits(student, target)
    (loss * scale).backward()
    return student.detach(), teacher, loss.detach(), {k: P[k].grad.float() / scale for k in pnames if P[k].grad is not None}


@pytest.mark.parametrize("variant", ["pyramid", "crossmodal"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_pyramid_vivit_matches_oracle(device, dtype, variant):
    """Toy size (64^2, T = 4, d = 128), EVERY parameter gradient.  fp32 kernels: 2e-3.  16-bit kernels: the protocol of
    tests/util.py -- within 2x the oracle's own low-precision deviation on the same inputs (larger of its autocast and
    its cast-to-16-bit run), per gradient and in the median.  BatchNorm on running statistics here (the 8-frame toy
    batch gives 2x2 maps at the deepest level); train-mode BatchNorm is exercised at full size below."""
    from dvt_amd.models.pyramid_vivit import PyramidViViT
    cm = variant == "crossmodal"
    torch.manual_seed(0)
    net = PyramidViViT(64, 19, 4, dim=128, depth=2, heads=2, dim_head=64, audio_tokens=32 if cm else 0, audio_dim=24,
                       distill=cm, compute_dtype=dtype)
    _randomise(net, 7)
    g = torch.Generator().manual_seed(8)
    clip = torch.randn(2, 4, 3, 64, 64, generator=g)
    audio = torch.randn(2, 32, 24, generator=g) if cm else None
    target = (torch.rand(2, 19, generator=g) < 0.3).float()
    net.eval()
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    pnames = [k for k, _ in net.named_parameters()]
    ref_s, ref_t, ref_loss, ref_g = _oracle_run(clip, audio, target, state, pnames, variant, 2, 2, None, False)
    net = net.cuda()
    batch = (target.cuda(), clip.cuda(), audio.cuda()) if cm else (target.cuda(), clip.cuda())
    scale = 1024.0 if dtype == torch.float16 else 1.0
    loss = net.training_step(batch)
    loss.backward(torch.tensor(scale, device="cuda"))
    Pn = dict(net.named_parameters())
    errs = {k: rel_l2(Pn[k].grad / scale, ref_g[k]) for k in ref_g}
    out = net(clip.cuda(), audio.cuda()) if cm else net(clip.cuda())
    e_out = rel_l2(out[0] if cm else out, ref_s)
    if cm:     # the teacher branch receives no gradient from the hard-label CE (argmax), like the reference
        assert Pn["distill_head.1.weight"].grad is None or float(Pn["distill_head.1.weight"].grad.abs().max()) == 0.0
    if dtype == torch.float32:
        assert abs(float(loss.detach()) - float(ref_loss)) < 2e-5 and e_out < 1e-4
        if cm:
            assert rel_l2(out[1], ref_t) < 1e-4
        for k, e in errs.items():
            assert e < 2e-3, (k, e)
        return
    prec = "bf16" if dtype == torch.bfloat16 else "fp16"
    yard, yard_out, yard_loss = {}, 0.0, 0.0
    for kind in ("amp", "pure"):
        s_, _, l_, g_ = _oracle_run(clip, audio, target, state, pnames, variant, 2, 2, f"{kind}_{prec}", False)
        yard_out = max(yard_out, rel_l2(s_, ref_s))
        yard_loss = max(yard_loss, abs(float(l_) - float(ref_loss)))
        for k in ref_g:
            yard[k] = max(yard.get(k, 0.0), rel_l2(g_[k], ref_g[k]))
    w = assert_within_reference_lowprec(f"toy {variant}/{prec}", e_out, errs, yard_out, yard)
    assert abs(float(loss.detach()) - float(ref_loss)) <= 2 * yard_loss + 2e-4
    print(f"[toy {variant}/{prec}] logits {e_out:.2e} (oracle's own {yard_out:.2e}); worst gradient ratio {w[0][1]:.2f} "
          f"({w[0][0]}); median ratio {w[1]:.2f}")


@pytest.mark.parametrize("variant", ["pyramid", "crossmodal"])
@pytest.mark.parametrize("mode", ["fp32", "bf16", "fp16"])
def test_pyramid_full_size_matches_digest(device, variant, mode):
    """BASELINE configs[2] / configs[3] at the size bench.py times them: one clip, T = 32, 224^2, d = 512, 4 + 4 layers,
    ResNet-18 pyramid under TRAIN-mode BatchNorm, 32 x 128 audio tokens + distillation head for configs[3].  Against
    tests/golden/pyramid_<variant>_digest.npz (tools/gen_golden.py pyramid_full: the oracle composition, whose stages
    are pinned to the imported reference one by one): logits, loss, and EVERY gradient's norm + 256 entries.
    fp32 kernels <= 1e-3 (north_star); bf16 / fp16 kernels within 2x the oracle's own low-precision deviation."""
    from dvt_amd.models.pyramid_vivit import PyramidViViT
    g = golden(f"pyramid_{variant}_digest.npz")
    cm = variant == "crossmodal"
    c = {k[4:]: int(g[k]) for k in g.files if k.startswith("cfg_")}
    dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[mode]
    net = PyramidViViT(c["image"], c["classes"], c["frames"], dim=c["dim"], depth=c["depth"], heads=c["heads"],
                       dim_head=c["dim_head"], audio_tokens=c["audio_tokens"], audio_dim=c["audio_dim"], distill=cm,
                       compute_dtype=dtype)
    fill_pyramid_from_numpy(net.named_parameters(), int(g["fill_seed"]))
    clip, audio, target = pyramid_digest_inputs(g)
    net = net.cuda().train()
    batch = (target.cuda(), clip.cuda(), audio.cuda()) if cm else (target.cuda(), clip.cuda())
    scale = 1024.0 if mode == "fp16" else 1.0
    loss = net.training_step(batch)
    loss.backward(torch.tensor(scale, device="cuda"))
    grads = {k: p.grad / scale for k, p in net.named_parameters() if p.grad is not None and "gn:" + k in g.files}
    assert set(grads) == {f[3:] for f in g.files if f.startswith("gn:")}          # every gradient of the digest is checked
    with torch.no_grad():      # logits: a second forward, still train mode (the same batch statistics)
        out = net(clip.cuda(), audio.cuda()) if cm else net(clip.cuda())
    e_out = rel_l2(out[0] if cm else out, torch.from_numpy(g["logits"]))
    e_loss = abs(float(loss.detach()) - float(g["loss"][0]))
    errs = grad_digest_errors(g, grads)
    wk = max(errs, key=errs.get)
    print(f"[full {variant}/{mode}] logits rel {e_out:.2e} loss abs {e_loss:.2e} worst grad digest {wk} {errs[wk]:.2e} "
          f"over {len(errs)} gradients")
    if mode == "fp32":
        # fp32 kernels against the oracle's FLOAT64 run, held to 2x the deviation of the oracle's own fp32 run from it (+ 1e-4):
        # behind 17 BatchNorm'd ReLU layers a gradient's fp32 noise floor is set by ReLU-mask flips of activations that are
        # 0 +- round-off (the oracle's fp32 backbone gradients deviate from its float64 ones by 1.5-3.7e-3 here; logits,
        # loss and every gradient outside the backbone by ~1e-6, and those are thereby held to 1e-4).
        t64 = _F64View(g)
        e64 = grad_digest_errors(t64, grads)
        own = grad_digest_errors(t64, {k[3:]: (g[k], g["gn:" + k[3:]]) for k in g.files if k.startswith("gs:")})
        wk = max(e64, key=lambda k: e64[k] / (own[k] + 1e-4))
        print(f"[full {variant}/fp32] vs float64: worst {wk} {e64[wk]:.2e} (oracle's own fp32 {own[wk]:.2e}); backbone median "
              f"{float(np.median([e for k, e in e64.items() if k.startswith('backbone')])):.2e} (oracle's own "
              f"{float(np.median([e for k, e in own.items() if k.startswith('backbone')])):.2e})")
        assert e_out < 1e-4 and e_loss < 1e-5
        if cm:
            assert rel_l2(out[1], torch.from_numpy(g["teacher"])) < 1e-4
        bb_med = float(np.median([e for k, e in own.items() if k.startswith("backbone")]))
        for k, e in e64.items():
            # a gradient's flip noise is a Poisson draw: a backbone parameter for which the oracle's fp32 run happened to
            # draw (almost) none is held to the backbone's typical (median) floor
            bound = 2 * max(own[k], bb_med) + 1e-4 if k.startswith("backbone") else 1e-4
            assert e <= bound, (k, e, own[k], bb_med)
        return
    ref_out, ref_errs = reference_lowprec_yardstick(g, g, mode)
    w = assert_within_reference_lowprec(f"full {variant}/{mode}", e_out, errs, ref_out, ref_errs, out_cap=3e-2)
    ref_loss = max(abs(float(g[f"{k}_{mode}:loss"][0]) - float(g["loss"][0])) for k in ("amp", "pure"))
    assert e_loss <= 2 * ref_loss + 2e-4, (e_loss, ref_loss)
    print(f"[full {variant}/{mode}] oracle's own {mode}: logits {ref_out:.2e}; worst gradient ratio {w[0][1]:.2f} ({w[0][0]}); "
          f"median ratio {w[1]:.2f}")
