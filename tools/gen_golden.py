#!/usr/bin/env python3
"""Generate golden vectors from the *imported reference* (build container only).

Run here (never on the GPU box -- /root/reference does not exist there):

    python tools/gen_golden.py

The reference modules are loaded by file path with importlib and are never
copied into this repository; only inputs / weights / outputs (data) are saved,
as small ``.npz`` files under ``tests/golden/``.

What is imported:
  * /root/reference/src/models/vit.py            (ViViT and its blocks)   -- as is
  * /root/reference/src/models/transformer.py    (PositionalEncoding)     -- needs
    ``pytorch_lightning``, which is not installed; a 6-line stand-in module whose
    ``LightningModule`` is ``torch.nn.Module`` is registered in ``sys.modules``
    for the duration of this script (generator only; nothing of it is shipped).
  * torch.nn.TransformerEncoderLayer from this container's torch: the
    reference's arithmetic for ``TransformerBase`` *is* that class
    (src/models/frame_transformer.py:41-44).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import digest_idx      # the one rule for which entries of a gradient a digest keeps  # noqa: E402

REF = "/root/reference/src/models"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEED = 1130  # src/main.py:25


def _load(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _np(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import fill_state_from_numpy, fill_resnet_from_numpy, fill_pyramid_from_numpy  # noqa: E402  (shared with the tests)


def fill_from_numpy(model: torch.nn.Module, seed: int) -> None:
    fill_state_from_numpy(model.named_parameters(), seed)


def vivit_case(vit, tag, cfg, batch, store_weights, seed):
    torch.manual_seed(seed)
    net = vit.ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["frames"], dim=cfg["dim"],
                    depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"])
    fill_from_numpy(net, seed + 1)
    rng = np.random.default_rng(seed + 2)
    x = torch.from_numpy(rng.standard_normal(
        (batch, cfg["frames"], 3, cfg["image"], cfg["image"])).astype(np.float32))
    y = torch.from_numpy((rng.random((batch, cfg["classes"])) < 0.2).astype(np.float32))
    y[:, 0] = 1.0
    logits = net(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, y)
    loss.backward()
    out = {"x": x.numpy(), "target": y.numpy(), "logits": logits.detach().numpy(),
           "loss": loss.detach().numpy()[None]}
    for k, v in cfg.items():
        out["cfg_" + k] = np.array(v)
    out["fill_seed"] = np.array(seed + 1)
    for name, p in net.named_parameters():
        if store_weights:
            out["w:" + name] = p.detach().numpy()
        out["g:" + name] = p.grad.detach().numpy()
    # intermediate activations for the tiny case (token assembly, space encoder)
    if store_weights:
        with torch.no_grad():
            e = net.to_patch_embedding(x)
            out["act:patch_embed"] = e.numpy()
    np.savez_compressed(os.path.join(OUT, f"vivit_{tag}.npz"), **out)
    print(f"vivit_{tag}: logits {tuple(logits.shape)} loss {loss.item():.6f}")


def block_cases(vit):
    torch.manual_seed(SEED)
    out = {}
    x = torch.randn(3, 10, 64)
    att = vit.Attention(64, heads=2, dim_head=32)
    ff = vit.FeedForward(64, 256)
    ln = torch.nn.LayerNorm(64)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(64)); ln.bias.copy_(0.1 * torch.randn(64))
    pre = vit.PreNorm(64, att)
    pre.norm = ln
    tr = vit.Transformer(64, 2, 2, 32, 128)
    xg = x.clone().requires_grad_(True)
    ya = att(xg); yf = ff(xg); yp = pre(xg); yt = tr(xg)
    out["x"] = x.numpy()
    out["attn_out"] = ya.detach().numpy(); out["ff_out"] = yf.detach().numpy()
    out["prenorm_attn_out"] = yp.detach().numpy(); out["tr_out"] = yt.detach().numpy()
    gy = torch.randn_like(yt)
    out["gy"] = gy.numpy()
    out["tr_gx"] = torch.autograd.grad((yt * gy).sum(), xg, retain_graph=True)[0].numpy()
    out["attn_gx"] = torch.autograd.grad((ya * gy).sum(), xg, retain_graph=True)[0].numpy()
    out["ff_gx"] = torch.autograd.grad((yf * gy).sum(), xg)[0].numpy()
    for k, v in att.state_dict().items():
        out["attn:" + k] = v.numpy()
    for k, v in ff.state_dict().items():
        out["ff:" + k] = v.numpy()
    out["ln:weight"] = ln.weight.detach().numpy(); out["ln:bias"] = ln.bias.detach().numpy()
    for k, v in tr.state_dict().items():
        out["tr:" + k] = v.numpy()
    # heads == 1 and dim_head == dim -> to_out is Identity (vit.py:34,41-44)
    att1 = vit.Attention(64, heads=1, dim_head=64)
    out["attn1_out"] = att1(x).detach().numpy()
    for k, v in att1.state_dict().items():
        out["attn1:" + k] = v.numpy()
    # patchify ordering (vit.py:90)
    from einops import rearrange
    img = torch.randn(2, 3, 3, 16, 24)
    out["patch_in"] = img.numpy()
    out["patch_out"] = rearrange(img, 'b t c (h p1) (w p2) -> b t (h w) (p1 p2 c)', p1=8, p2=8).numpy()
    np.savez_compressed(os.path.join(OUT, "vit_blocks.npz"), **out)
    print("vit_blocks: ok")


def encoder_layer_case():
    """torch's own TransformerEncoderLayer == the arithmetic of TransformerBase
    (frame_transformer.py:41-44): post-norm, ReLU, seq-first, eval mode."""
    torch.manual_seed(SEED)
    d, nhead, ff, L, B = 64, 2, 96, 7, 3
    layer = torch.nn.TransformerEncoderLayer(d, nhead, ff, 0.5)
    enc = torch.nn.TransformerEncoder(layer, 2, enable_nested_tensor=False)
    enc.eval()
    x = torch.randn(L, B, d, requires_grad=True)
    y = enc(x)
    gy = torch.randn_like(y)
    gx = torch.autograd.grad((y * gy).sum(), x)[0]
    out = {"x": x.detach().numpy(), "y": y.detach().numpy(), "gy": gy.numpy(), "gx": gx.numpy(),
           "nhead": np.array(nhead)}
    for k, v in enc.state_dict().items():
        out["w:transformer." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "encoder_postnorm.npz"), **out)
    print("encoder_postnorm: ok")


def posenc_case():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):  # generator-only stand-in
        pass

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    try:
        tr = _load("ref_transformer", os.path.join(REF, "transformer.py"))
    finally:
        del sys.modules["pytorch_lightning"]
    out = {}
    for d, L in ((896, 14), (2048, 14), (64, 9)):
        pe = tr.PositionalEncoding(d, 0.0, max_len=L)
        x = torch.zeros(L, 2, d)
        out[f"pe_{d}_{L}"] = pe.pe.numpy()
        out[f"fwd_{d}_{L}"] = pe.eval()(x + 1.0).numpy()
    np.savez_compressed(os.path.join(OUT, "posenc.npz"), **out)
    print("posenc: ok")


def resnet_case():
    """Reference custom_resnet.resnet18 at 224x224 (the only size its fixed AvgPool2d(7) admits),
    train mode (BatchNorm batch statistics), batch 2.  Weights/inputs come from numpy streams and are
    not stored; stored: the pyramid (x2, x3, x4), a few small gradients, gradient norms of every
    parameter, and two updated running statistics."""
    cr = _load("ref_custom_resnet", os.path.join(REF, "custom_resnet.py"))
    net = cr.resnet18(False)
    rng = np.random.default_rng(SEED + 20)
    fill_resnet_from_numpy(net, rng)
    net.train()
    x = torch.from_numpy(rng.standard_normal((2, 3, 224, 224)).astype(np.float32))
    x2, x3, x4 = net(x)
    gs = [torch.from_numpy(rng.standard_normal(tuple(t.shape)).astype(np.float32)) for t in (x2, x3, x4)]
    loss = sum((t * g).sum() for t, g in zip((x2, x3, x4), gs)) / 1000.0
    loss.backward()
    out = {"seed": np.array(SEED + 20), "x2": x2.detach().numpy(), "x3": x3.detach().numpy(),
           "x4": x4.detach().numpy(), "loss": loss.detach().numpy()[None]}
    keep = ["conv1.weight", "bn1.weight", "bn1.bias", "layer1.0.conv1.weight", "layer2.0.downsample.0.weight",
            "layer2.0.downsample.1.weight", "layer4.1.bn2.weight", "layer4.1.bn2.bias", "layer3.1.conv2.weight"]
    names, norms = [], []
    for name, p in net.named_parameters():
        if p.grad is None:
            continue
        names.append(name)
        norms.append(float(p.grad.double().norm()))
        if name in keep and p.grad.numel() < 700000:
            out["g:" + name] = p.grad.numpy()
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms)
    out["rm:bn1"] = net.bn1.running_mean.numpy(); out["rv:bn1"] = net.bn1.running_var.numpy()
    out["rm:layer4.1.bn2"] = net.layer4[1].bn2.running_mean.numpy()
    out["rv:layer4.1.bn2"] = net.layer4[1].bn2.running_var.numpy()
    np.savez_compressed(os.path.join(OUT, "resnet18_pyramid.npz"), **out)
    print("resnet18_pyramid: x2", tuple(x2.shape), "x3", tuple(x3.shape), "x4", tuple(x4.shape), "loss", float(loss))


def resnet_lowprec_case():
    """The reference ResNet-18's OWN low-precision deviation on the ``resnet18_pyramid`` fixture inputs (torch.autocast on
    the CPU, train-mode BatchNorm): relative L2 error of (x2, x3, x4) and of every parameter gradient against the
    reference's fp32 run, plus the gradient cosines.  Stored as scalars; the GPU test holds the bf16 / fp16 kernels to
    a multiple of these.  (Measured here: the reference's own bf16 gradients deviate by 0.24-0.45 with cosines down to
    0.90 on this fixture, at batch 2 and at batch 16 alike -- 17 BatchNorm'd ReLU layers under a random output
    gradient are ill-conditioned in 8-bit mantissas whoever computes them.)"""
    cr = _load("ref_custom_resnet3", os.path.join(REF, "custom_resnet.py"))

    def run(dt):
        net = cr.resnet18(False)
        rng = np.random.default_rng(SEED + 20)
        fill_resnet_from_numpy(net, rng)
        net.train()
        x = torch.from_numpy(rng.standard_normal((2, 3, 224, 224)).astype(np.float32))
        if dt is None:
            outs = net(x)
        else:
            with torch.autocast("cpu", dtype=dt):
                outs = net(x)
        gs = [torch.from_numpy(rng.standard_normal(tuple(t.shape)).astype(np.float32)) for t in outs]
        scale = 1024.0 if dt == torch.float16 else 1.0
        loss = sum((t.float() * g).sum() for t, g in zip(outs, gs)) / 1000.0
        (loss * scale).backward()
        return ([o.detach().double() for o in outs],
                {k: p.grad.double() / scale for k, p in net.named_parameters() if p.grad is not None})

    ref_o, ref_g = run(None)
    out = {"names": np.array(list(ref_g))}
    for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        o, g = run(dt)
        out[f"{tag}:out_err"] = np.array([float((a - b).norm() / b.norm()) for a, b in zip(o, ref_o)])
        out[f"{tag}:grad_err"] = np.array([float((g[k] - ref_g[k]).norm() / ref_g[k].norm()) for k in ref_g])
        out[f"{tag}:grad_cos"] = np.array([float(g[k].flatten() @ ref_g[k].flatten() / (g[k].norm() * ref_g[k].norm()))
                                           for k in ref_g])
        print(f"resnet18_lowprec[{tag}]: out err {out[tag + ':out_err']}, grad err median "
              f"{np.median(out[tag + ':grad_err']):.3f} max {out[tag + ':grad_err'].max():.3f}, "
              f"min cos {out[tag + ':grad_cos'].min():.4f}")
    np.savez_compressed(os.path.join(OUT, "resnet18_lowprec.npz"), **out)


def tpn_case():
    """src/models/TPN.py has no import statements (NameError on import).  Its source is executed here
    with the missing names supplied (torch, nn, a LightningModule stand-in, the reference's own
    custom_resnet) so that Reasoning / sum_group / Feature_Pyramid_* run as written.  TPN() itself is not
    instantiated (it downloads pretrained weights)."""
    class _PL:
        LightningModule = torch.nn.Module
    ns = {"torch": torch, "nn": torch.nn, "pl": _PL, "custom_resnet": _load("ref_custom_resnet2", os.path.join(REF, "custom_resnet.py"))}
    exec(compile(open(os.path.join(REF, "TPN.py")).read(), "TPN.py", "exec"), ns)
    rng = np.random.default_rng(SEED + 30)
    out = {"seed": np.array(SEED + 30)}
    reason = ns["Reasoning"]().eval()
    fill_resnet_from_numpy(reason, rng)                    # Linear weights 0.02 n, biases 0.1 n
    x = torch.from_numpy(rng.standard_normal((1, 20, 896)).astype(np.float32)).requires_grad_(True)
    y = reason(x)
    gy = torch.from_numpy(rng.standard_normal((1, 15)).astype(np.float32))
    (y * gy).sum().backward()
    out["reason_out"] = y.detach().numpy(); out["reason_gx"] = x.grad.numpy()
    out["reason_gw_last"] = reason.relation[2][7].weight.grad.numpy()
    out["sum_group3"] = ns["sum_group"](x.detach(), 3).numpy()
    for name, cls, shape in (("low", "Feature_Pyramid_low", (3, 128, 28, 28)), ("mid", "Feature_Pyramid_Mid", (3, 256, 14, 14)),
                             ("high", "Feature_Pyramid_High", (3, 512, 7, 7))):
        m = ns[cls]().eval()
        fill_resnet_from_numpy(m, rng)
        f = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
        out["pyr_" + name] = m(f).squeeze().detach().numpy()
    np.savez_compressed(os.path.join(OUT, "tpn_pieces.npz"), **out)
    print("tpn_pieces: reasoning out", tuple(y.shape))


def vivit_digest_case(vit, tag, cfg, batch, seed):
    """Large configurations (BASELINE configs[1] and the metric shape): the clip and the weights are
    regenerated from numpy seeds by the tests, so only the reference's outputs are stored -- logits, loss
    and, per parameter gradient, its L2 norm plus 256 evenly spaced entries."""
    torch.manual_seed(seed)
    net = vit.ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["frames"], dim=cfg["dim"],
                    depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"])
    fill_from_numpy(net, seed + 1)
    rng = np.random.default_rng(seed + 2)
    x = torch.from_numpy(rng.standard_normal(
        (batch, cfg["frames"], 3, cfg["image"], cfg["image"])).astype(np.float32))
    y = torch.from_numpy((rng.random((batch, cfg["classes"])) < 0.2).astype(np.float32))
    y[:, 0] = 1.0
    logits = net(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, y)
    loss.backward()
    out = {"target": y.numpy(), "logits": logits.detach().numpy(), "loss": loss.detach().numpy()[None],
           "fill_seed": np.array(seed + 1), "x_seed": np.array(seed + 2), "batch": np.array(batch)}
    for k, v in cfg.items():
        out["cfg_" + k] = np.array(v)
    for name, p in net.named_parameters():
        idx = digest_idx(tuple(p.grad.shape))            # tests/util.py: 256 spaced entries; pos_embedding: + its CLS rows
        g = p.grad.detach().reshape(-1)
        out["gn:" + name] = np.array(float(g.double().norm()))
        out["gs:" + name] = g[torch.from_numpy(idx)].numpy()
    np.savez_compressed(os.path.join(OUT, f"vivit_{tag}.npz"), **out)
    print(f"vivit_{tag}: logits {tuple(logits.shape)} loss {loss.item():.6f}")


def _lowprec_run(vit, cfg, batch, seed, mode):
    """The reference's OWN low-precision CPU run on the digest inputs: ``amp_*`` = torch.autocast on the CPU (what
    Lightning's ``precision=16`` in the comment at src/main.py:85 would do; reproduces BASELINE.md section 2's
    4.6e-3 / 1e-2 at configs[0]); ``pure_*`` = module and clip cast to the 16-bit type.  fp16 runs use a static loss
    scale of 1024 (activation gradients underflow without one), removed from the gradients afterwards."""
    torch.manual_seed(seed)
    net = vit.ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["frames"], dim=cfg["dim"],
                    depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"])
    fill_from_numpy(net, seed + 1)
    rng = np.random.default_rng(seed + 2)
    x = torch.from_numpy(rng.standard_normal(
        (batch, cfg["frames"], 3, cfg["image"], cfg["image"])).astype(np.float32))
    y = torch.from_numpy((rng.random((batch, cfg["classes"])) < 0.2).astype(np.float32))
    y[:, 0] = 1.0
    kind, prec = mode.split("_")
    dt = torch.bfloat16 if prec == "bf16" else torch.float16
    scale = 1024.0 if prec == "fp16" else 1.0
    if kind == "amp":
        with torch.autocast("cpu", dtype=dt):
            logits = net(x)
    else:
        net = net.to(dt)
        logits = net(x.to(dt))
    loss = torch.nn.BCEWithLogitsLoss()(logits.float(), y)
    (loss * scale).backward()
    return logits.detach().float(), loss.detach(), {k: p.grad.detach().float() / scale for k, p in net.named_parameters()}


def vivit_lowprec_case(vit, tag, cfg, batch, seed, modes):
    """``vivit_<tag>_lowprec.npz``: digests (logits, loss, per-gradient norm + the same 256 evenly spaced entries as
    the fp32 digest) of the reference's own low-precision runs, so that the GPU tests can hold the bf16 / fp16 kernels
    to "<= 2x the reference's own low-precision deviation on the same inputs" (BASELINE.md section 2)."""
    out = {"modes": np.array(modes), "batch": np.array(batch), "fill_seed": np.array(seed + 1), "x_seed": np.array(seed + 2)}
    for mode in modes:
        logits, loss, grads = _lowprec_run(vit, cfg, batch, seed, mode)
        out[f"{mode}:logits"] = logits.numpy()
        out[f"{mode}:loss"] = loss.numpy()[None]
        for name, g in grads.items():
            idx = digest_idx(tuple(g.shape))
            g = g.reshape(-1)
            out[f"{mode}:gn:{name}"] = np.array(float(g.double().norm()))
            out[f"{mode}:gs:{name}"] = g[torch.from_numpy(idx)].numpy()
        print(f"vivit_{tag}_lowprec[{mode}]: loss {float(loss):.6f}")
    np.savez_compressed(os.path.join(OUT, f"vivit_{tag}_lowprec.npz"), **out)


def input_stage_case():
    """Resize + CenterCrop + ToTensor + Normalize (MMX_Light_dl.py:203-217) through Pillow itself (the resize is
    Pillow arithmetic) and torch (ToTensor / Normalize restated: torchvision is not installed)."""
    from PIL import Image
    rng = np.random.default_rng(SEED + 40)
    out = {}
    mean, std = (0.43216, 0.394666, 0.37645), (0.22803, 0.22145, 0.216989)
    cases = [("down_wide", (3, 90, 160), 40, 32), ("down_tall", (2, 150, 84), 36, 36), ("up", (2, 48, 64), 56, 56),
             ("train_vid", (1, 135, 240), 120, 112), ("identity", (1, 32, 32), 32, 32)]
    for tag, (F_, H0, W0), resize, crop in cases:
        noise = rng.integers(0, 256, (F_, H0, W0, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:H0, 0:W0]
        smooth = np.stack([(yy * 255 // max(H0 - 1, 1)), (xx * 255 // max(W0 - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
        frames = noise.copy()
        frames[0] = smooth if F_ > 1 else noise[0]
        res = []
        for f in range(F_):
            img = Image.fromarray(frames[f])
            w, h = img.size
            if w <= h:
                ow, oh = resize, int(resize * h / w)
            else:
                oh, ow = resize, int(resize * w / h)
            img = img.resize((ow, oh), Image.BILINEAR)
            top, left = int(round((oh - crop) / 2.0)), int(round((ow - crop) / 2.0))
            img = img.crop((left, top, left + crop, top + crop))
            t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).to(torch.float32).div(255)
            t = (t - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
            res.append(t.numpy())
        out[f"{tag}:frames"] = frames
        out[f"{tag}:out"] = np.stack(res)
        out[f"{tag}:cfg"] = np.array([resize, crop])
    out["mean"], out["std"] = np.array(mean), np.array(std)
    np.savez_compressed(os.path.join(OUT, "input_stage.npz"), **out)
    print("input_stage:", [c[0] for c in cases])


def eval_metrics_case():
    """callbacks.py:36-55 through scikit-learn itself (the reference's dependency)."""
    import warnings
    from sklearn.metrics import average_precision_score, f1_score
    rng = np.random.default_rng(SEED + 50)
    N, C = 300, 19
    y = (rng.random((N, C)) < 0.2).astype(np.uint8)
    y[:, 14] = 0                                    # a class without positives (TVMovie is rare)
    y[5, :] = 0                                     # a sample without positives
    s = rng.random((N, C)).astype(np.float32)
    s[:, 7] = np.round(s[:, 7], 1)                  # heavy ties
    s[::9, 2] = 0.5
    t = [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = {"probs": s, "labels": y, "thresholds": np.array(t),
               "f1": np.array([f1_score(y, (s > tt).astype(int), average="samples", zero_division=0) for tt in t]),
               "ap_samples": np.array(average_precision_score(y, s, average="samples")),
               "ap_weighted": np.array(average_precision_score(y, s, average="weighted")),
               "ap_class": np.array(average_precision_score(y, s, average=None))}
    np.savez_compressed(os.path.join(OUT, "eval_metrics.npz"), **out)
    print("eval_metrics: ap_samples", float(out["ap_samples"]), "weighted", float(out["ap_weighted"]))


def fusion_case():
    """ContrastiveLoss from the imported reference (losses/ntxent.py) and CollaborativeGating from the executed
    reference text (collabgating.py has no imports: torch / nn / F / a LightningModule stand-in are supplied)."""
    import torch.nn.functional as TF
    nt = _load("ref_ntxent", os.path.join(REF, "losses", "ntxent.py"))
    rng = np.random.default_rng(SEED + 60)
    out = {}
    B, D = 6, 40
    zi = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).requires_grad_(True)
    zj = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).requires_grad_(True)
    loss = nt.ContrastiveLoss(B, 0.5)(zi, zj)
    loss.backward()
    out.update({"cl_zi": zi.detach().numpy(), "cl_zj": zj.detach().numpy(), "cl_loss": loss.detach().numpy()[None],
                "cl_gzi": zi.grad.numpy(), "cl_gzj": zj.grad.numpy()})

    class _PL:
        LightningModule = torch.nn.Module
    ns = {"torch": torch, "nn": torch.nn, "F": TF, "pl": _PL}
    exec(compile(open(os.path.join(REF, "collabgating.py")).read(), "collabgating.py", "exec"), ns)
    cg = ns["CollaborativeGating"]().eval()
    wrng = np.random.default_rng(SEED + 61)                    # weights are regenerated by the tests from this seed
    with torch.no_grad():
        for name, p in cg.named_parameters():
            a = wrng.standard_normal(tuple(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(a * np.float32(0.02 if p.dim() == 2 else 0.1)))
    out["cg_wseed"] = np.array(SEED + 61)
    out["cg_wnames"] = np.array([n for n, _ in cg.named_parameters()])
    dims = (2048, 2048, 128)                                   # the third expert is stretched to 2048
    feats = [[[torch.from_numpy(rng.standard_normal((1, d)).astype(np.float32)) for d in dims] for _ in range(2)]
             for _ in range(2)]
    for bi, scenes in enumerate(feats):
        for si, experts in enumerate(scenes):
            for ei, t in enumerate(experts):
                out[f"cg_x:{bi}:{si}:{ei}"] = t.numpy().copy()
    y = cg([[list(e) for e in scenes] for scenes in feats])    # the reference mutates the expert lists
    gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    (y * gy).sum().backward()
    out["cg_out"], out["cg_gy"] = y.detach().numpy(), gy.numpy()
    for name, p in cg.named_parameters():
        g = p.grad.detach().reshape(-1)
        idx = np.linspace(0, g.numel() - 1, num=min(256, g.numel())).astype(np.int64)
        out["gn:" + name] = np.array(float(g.double().norm()))
        out["gs:" + name] = g[torch.from_numpy(idx)].numpy()
    np.savez_compressed(os.path.join(OUT, "fusion.npz"), **out)
    print("fusion: contrastive loss", float(loss.detach()), "gating out", tuple(y.shape))


def pyramid_digest_case(variant, small=False):
    """BASELINE configs[2] ('pyramid') / configs[3] ('crossmodal') at FULL size -- one clip, T = 32, 224^2, d = 512,
    4 + 4 layers, 8 heads, ResNet-18 pyramid with TRAIN-mode BatchNorm (batch statistics over the 32 frames), 32 audio
    tokens of width 128 and the distillation head for configs[3] -- through ``oracle.pyramid_path`` (every stage of it
    is pinned to the imported reference by its own fixture; the wiring is build-defined, models/pyramid_vivit.py).
    Stored: logits, loss, per gradient its L2 norm + 256 evenly spaced entries -- for the fp32 run and for the oracle's
    own low-precision runs (``amp_*``: torch.autocast; ``pure_*``: parameters and inputs cast to the 16-bit type), the
    yardsticks of the 16-bit protocol.  Weights / inputs are regenerated by the tests from the stored numpy seeds; the
    build's module is instantiated here only as the container that names and shapes the parameters."""
    from dvt_amd.models.pyramid_vivit import PyramidViViT
    from oracle import pyramid_path as PP
    from oracle import clip_path as O
    cm = variant == "crossmodal"
    seed = SEED + (90 if cm else 80)
    cfg = dict(image=224, frames=32, dim=512, depth=4, heads=8, dim_head=64, classes=19, audio_tokens=32 if cm else 0,
               audio_dim=128)
    if small:                                  # generator self-check size
        cfg.update(image=64, frames=4, dim=128, depth=2, heads=2)
    net = PyramidViViT(cfg["image"], cfg["classes"], cfg["frames"], dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"],
                       dim_head=cfg["dim_head"], audio_tokens=cfg["audio_tokens"], audio_dim=cfg["audio_dim"], distill=cm,
                       compute_dtype=torch.float32)
    fill_pyramid_from_numpy(net.named_parameters(), seed + 1)
    rng = np.random.default_rng(seed + 2)
    clip = torch.from_numpy(rng.standard_normal((1, cfg["frames"], 3, cfg["image"], cfg["image"])).astype(np.float32))
    audio = torch.from_numpy(rng.standard_normal((1, 32, 128)).astype(np.float32)) if cm else None
    y = torch.from_numpy((rng.random((1, cfg["classes"])) < 0.2).astype(np.float32))
    y[:, 0] = 1.0
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    pnames = [k for k, _ in net.named_parameters()]

    def run(mode):
        kind, prec = (mode.split("_") + [None])[:2] if mode else (None, None)
        if mode == "f64":                      # the arithmetic's own noise floor: fp32 run vs this one
            kind, prec = "pure", "f64"
        dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "f64": torch.float64, None: torch.float32}[prec]
        scale = 1024.0 if prec == "fp16" else 1.0
        cast = (lambda t: t.to(dt)) if kind == "pure" else (lambda t: t)
        P = {k: (cast(v) if v.dtype.is_floating_point else v).clone() for k, v in state.items()}
        for k in pnames:
            P[k].requires_grad_(True)
        xin, ain = cast(clip), (cast(audio) if cm else None)

        def fwd():
            return PP.pyramid_vivit_forward(xin, ain, P, depth=cfg["depth"], heads=cfg["heads"], training_bn=True, distill=cm)

        if kind == "amp":
            with torch.autocast("cpu", dtype=dt):
                out = fwd()
        else:
            out = fwd()
        if cm:
            student, teacher = out[0].float(), out[1].float()
            loss = O.bce_with_logits(student, y) + O.cross_entropy_hard(student, teacher)
        else:
            student, teacher = out.float(), None
            loss = O.bce_with_logits(student, y)
        (loss * scale).backward()
        grads = {k: (P[k].grad.double() / scale if P[k].grad is not None else None) for k in pnames}
        return student.detach(), (teacher.detach() if cm else None), loss.detach(), grads

    out = {"fill_seed": np.array(seed + 1), "x_seed": np.array(seed + 2), "batch": np.array(1), "target": y.numpy()}
    for k, v in cfg.items():
        out["cfg_" + k] = np.array(v)
    path = os.path.join(OUT, f"pyramid_{variant}_digest.npz")
    have = {}
    if not small and os.path.exists(path) and "--force" not in sys.argv:       # keep the modes already generated
        have = dict(np.load(path))
        if all(int(have.get("cfg_" + k, -1)) == v for k, v in cfg.items()) and int(have["fill_seed"]) == seed + 1:
            out.update(have)
        else:
            have = {}
    for mode in (None, "f64", "amp_bf16", "pure_bf16", "amp_fp16", "pure_fp16"):
        pre = (mode + ":") if mode else ""
        if pre + "logits" in have:
            continue
        student, teacher, loss, grads = run(mode)
        out[pre + "logits"] = student.numpy()
        if cm:
            out[pre + "teacher"] = teacher.numpy()
        out[pre + "loss"] = loss.numpy()[None]
        for name, g in grads.items():
            if g is None:                          # distill_head: the hard-label CE passes no gradient to the teacher
                continue
            g = g.reshape(-1)
            idx = np.linspace(0, g.numel() - 1, num=min(256, g.numel())).astype(np.int64)
            out[f"{pre}gn:{name}"] = np.array(float(g.double().norm()))
            out[f"{pre}gs:{name}"] = g[torch.from_numpy(idx)].numpy().astype(np.float64 if mode == "f64" else np.float32)
        print(f"pyramid_{variant}_digest[{mode or 'fp32'}]: loss {float(loss):.6f}", flush=True)
    if not small:
        np.savez_compressed(path, **out)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = set(a for a in sys.argv[1:] if not a.startswith("--"))                 # e.g. ``gen_golden.py lowprec longclip`` regenerates just those groups

    def want(group):
        return not only or group in only

    vit = _load("ref_vit", os.path.join(REF, "vit.py"))
    tiny = dict(image=32, patch=8, classes=19, frames=3, dim=64, depth=2, heads=2, dim_head=32)
    # BASELINE.json configs[0]: T=4, 64x64, d=128, 2 layers (plumbing config)
    c1 = dict(image=64, patch=16, classes=19, frames=4, dim=128, depth=2, heads=2, dim_head=64)
    # BASELINE.json configs[1]: single-modal video path d=384, T=16, 224^2 (one clip); and the metric shape
    c2 = dict(image=224, patch=16, classes=19, frames=16, dim=384, depth=4, heads=6, dim_head=64)
    cm = dict(image=224, patch=16, classes=19, frames=32, dim=512, depth=4, heads=8, dim_head=64)
    # BASELINE.json configs[4]: long clip, T=64, 288^2 (N = 325 tokens per frame), one clip
    c5 = dict(image=288, patch=16, classes=19, frames=64, dim=512, depth=4, heads=8, dim_head=64)
    if want("vivit"):
        vivit_case(vit, "tiny", tiny, batch=2, store_weights=True, seed=SEED)
        vivit_case(vit, "c1", c1, batch=2, store_weights=False, seed=SEED + 10)
        vivit_digest_case(vit, "c2_digest", c2, batch=1, seed=SEED + 20)
        vivit_digest_case(vit, "metric_digest", cm, batch=1, seed=SEED + 30)
    if want("longclip"):
        vivit_digest_case(vit, "longclip_digest", c5, batch=1, seed=SEED + 70)
    if want("lowprec"):
        # amp_* = torch.autocast (residual stream and LayerNorm in fp32), pure_* = module and clip cast to the 16-bit
        # type (residual stream stored in 16 bits, as the HIP path stores it): the protocol's yardstick per quantity
        # is the larger of the two (tests/util.py)
        modes = ["amp_bf16", "pure_bf16", "amp_fp16", "pure_fp16"]
        vivit_lowprec_case(vit, "tiny", tiny, 2, SEED, modes)
        vivit_lowprec_case(vit, "c1", c1, 2, SEED + 10, modes)
        vivit_lowprec_case(vit, "c2", c2, 1, SEED + 20, modes)
        vivit_lowprec_case(vit, "metric", cm, 1, SEED + 30, modes)
        vivit_lowprec_case(vit, "longclip", c5, 1, SEED + 70, modes)
    if want("metric_b8"):
        # the batch bench.py TIMES (B = 8 per GPU at the metric shape): same digest form, so the GPU tests hold the HIP
        # path to the executed reference at the exact grid sizes of the headline
        vivit_digest_case(vit, "metric_b8_digest", cm, batch=8, seed=SEED + 80)
        vivit_lowprec_case(vit, "metric_b8", cm, 8, SEED + 80, modes=["amp_bf16", "pure_bf16", "amp_fp16", "pure_fp16"])
    if want("pyramid_full"):      # existing modes of an existing fixture are kept unless --force is given
        pyramid_digest_case("pyramid")
        pyramid_digest_case("crossmodal")
    for group, fn in (("blocks", lambda: block_cases(vit)), ("encoder", encoder_layer_case), ("posenc", posenc_case),
                      ("resnet", resnet_case), ("resnet_lowprec", resnet_lowprec_case), ("tpn", tpn_case), ("input_stage", input_stage_case),
                      ("eval_metrics", eval_metrics_case), ("fusion", fusion_case)):
        if want(group):
            fn()


if __name__ == "__main__":
    main()
