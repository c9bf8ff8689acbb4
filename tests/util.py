"""Shared test helpers (no reference code; data plumbing only)."""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name))


def fill_state_from_numpy(named_params, seed: int) -> None:
    """Deterministic, platform-independent parameter fill (numpy PCG64 stream),
    consumed in ``named_parameters()`` order.  Used by tools/gen_golden.py on the
    reference model and by the tests on the build's mirror, so large fixtures
    do not have to store weights.

    1-D ``*.weight`` (LayerNorm scale) -> 1 + 0.1 n;  1-D other (biases) -> 0.1 n;
    tokens / positional table -> 0.5 n;  matrices -> 0.05 n.
    """
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in named_params:
            a = rng.standard_normal(tuple(p.shape)).astype(np.float32)
            if p.dim() == 1:
                a = (1.0 + 0.1 * a) if name.endswith("weight") else 0.1 * a
            elif "token" in name or "pos_embedding" in name:
                a = 0.5 * a
            else:
                a = 0.05 * a
            p.copy_(torch.from_numpy(a).to(p.dtype))


def fill_resnet_from_numpy(net, rng) -> None:
    """Parameter fill shared by tools/gen_golden.py (reference ResNet) and the tests (the build's
    mirror): He-scaled conv weights, 0.02 n for fc, 1 + 0.1 n BN scales, 0.1 n shifts, consumed in
    ``named_parameters()`` order from one numpy Generator."""
    with torch.no_grad():
        for name, p in net.named_parameters():
            a = rng.standard_normal(tuple(p.shape)).astype(np.float32)
            if p.dim() == 4:
                a *= np.float32(np.sqrt(2.0 / (p.shape[0] * p.shape[2] * p.shape[3])))
            elif p.dim() == 2:
                a *= np.float32(0.02)
            elif name.endswith("weight"):
                a = 1 + np.float32(0.1) * a
            else:
                a = np.float32(0.1) * a
            p.copy_(torch.from_numpy(a).to(p.dtype))


def fill_pyramid_from_numpy(named_params, seed: int) -> None:
    """Parameter fill of the composed configs[2] / configs[3] model (shared by tools/gen_golden.py and the tests, so the
    full-size fixtures store no weights): convolutions He-scaled, matrices n / sqrt(fan_in), LayerNorm / BatchNorm
    scales 1 + 0.1 n, biases 0.1 n, tokens and the positional table 0.5 n -- one numpy PCG64 stream consumed in
    ``named_parameters()`` order."""
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in named_params:
            a = rng.standard_normal(tuple(p.shape)).astype(np.float32)
            if "token" in name or "pos_embedding" in name:
                a = np.float32(0.5) * a
            elif p.dim() == 4:
                a *= np.float32(np.sqrt(2.0 / (p.shape[1] * p.shape[2] * p.shape[3])))
            elif p.dim() == 2:
                a *= np.float32(1.0 / np.sqrt(p.shape[1]))
            elif name.endswith("weight"):
                a = 1 + np.float32(0.1) * a
            else:
                a = np.float32(0.1) * a
            p.copy_(torch.from_numpy(a).to(p.dtype))


def pyramid_digest_inputs(npz):
    """Clip / audio / target of a ``pyramid_*_digest.npz`` fixture, regenerated from its numpy seed."""
    rng = np.random.default_rng(int(npz["x_seed"]))
    b, t, image = int(npz["batch"]), int(npz["cfg_frames"]), int(npz["cfg_image"])
    clip = torch.from_numpy(rng.standard_normal((b, t, 3, image, image)).astype(np.float32))
    audio = None
    if int(npz["cfg_audio_tokens"]):
        audio = torch.from_numpy(rng.standard_normal((b, int(npz["cfg_audio_tokens"]), int(npz["cfg_audio_dim"]))).astype(np.float32))
    return clip, audio, torch.from_numpy(npz["target"])


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def digest_inputs(npz):
    """Clip regenerated from the numpy seed stored in a ``vivit_*_digest.npz`` fixture (the generator drew it
    from the same PCG64 stream)."""
    cfg = {k[4:]: int(npz[k]) for k in npz.files if k.startswith("cfg_")}
    rng = np.random.default_rng(int(npz["x_seed"]))
    x = rng.standard_normal((int(npz["batch"]), cfg["frames"], 3, cfg["image"], cfg["image"])).astype(np.float32)
    return cfg, torch.from_numpy(x), torch.from_numpy(npz["target"])


def check_grad_digest(npz, grads, tol, label=""):
    """grads: name -> tensor.  Every gradient's L2 norm and its 256 evenly spaced entries against the digest."""
    worst = ("", 0.0)
    for k in [f[3:] for f in npz.files if f.startswith("gs:")]:
        g = grads[k].detach().float().cpu().reshape(-1)
        ref_s, ref_n = torch.from_numpy(npz["gs:" + k]), float(npz["gn:" + k])
        idx = digest_idx(tuple(grads[k].shape), stored=ref_s.numel())
        e_s = float((g[torch.from_numpy(idx)] - ref_s).norm() / (ref_s.norm() + 1e-30))
        e_n = abs(float(g.double().norm()) - ref_n) / (ref_n + 1e-30)
        e = max(e_s, e_n)
        if e > worst[1]:
            worst = (k, e)
        assert e < tol, (label, k, e_s, e_n)
    return worst


def digest_idx(shape, stored=None):
    """Flat indices of the entries a gradient digest keeps (the SAME rule in tools/gen_golden.py and in the checks).
    Default: 256 evenly spaced entries.  A 4-D gradient [1, T, rows, d] -- ``pos_embedding`` -- is, at one clip, the raw
    gradient of the token map, and its energy sits in the T CLS rows (row 0 of every frame: > 95 %; the patch rows only
    receive what attention routes back from a CLS query): 256 evenly spaced entries catch one or two CLS-row entries, so
    the digest error of that one parameter was a coin toss (ratios between 0.9 and 4.0 from run to run on the same
    kernels).  Its digest therefore holds all T * d CLS-row entries followed by the 256 spaced ones.  ``stored``: the
    entry count of an existing fixture -- older ones (256 entries for every tensor) keep their rule."""
    n = int(np.prod(shape))
    base = np.linspace(0, n - 1, num=min(256, n)).astype(np.int64)
    if len(shape) != 4 or shape[0] != 1 or (stored is not None and stored == base.size):
        return base
    _, T, R, d = shape
    cls = (np.arange(T)[:, None] * (R * d) + np.arange(d)[None, :]).reshape(-1).astype(np.int64)
    return np.concatenate([cls, base])


def _digest_idx(n):
    return digest_idx((n,))


def grad_digest_errors(npz, grads, prefix=""):
    """name -> max(rel-L2 error on the 256 digest entries, rel error of the L2 norm) of ``grads`` (name -> tensor or
    flat numpy array of the digest entries when ``prefix`` names a low-precision run stored in digest form) against the
    fp32 digest / full gradients held by ``npz``."""
    out = {}
    names = [f[3:] for f in npz.files if f.startswith("gs:")] or [f[2:] for f in npz.files if f.startswith("g:")]
    for k in names:
        if "gs:" + k in npz.files:
            ref_s, ref_n = torch.from_numpy(npz["gs:" + k]).double(), float(npz["gn:" + k])
        else:
            full = torch.from_numpy(npz["g:" + k]).double()
            stored_lp = grads[k][0].size if isinstance(grads[k], tuple) else None      # a stored low-precision digest decides the rule
            ref_s, ref_n = full.reshape(-1)[torch.from_numpy(digest_idx(tuple(full.shape), stored=stored_lp))], float(full.norm())
        got = grads[k]
        if isinstance(got, tuple):                       # (digest entries, norm) of a stored low-precision run
            g_s, g_n = torch.from_numpy(got[0]).double(), float(got[1])
        else:
            g = got.detach().double().cpu().reshape(-1)
            g_s, g_n = g[torch.from_numpy(digest_idx(tuple(got.shape), stored=ref_s.numel()))], float(g.norm())
        e_s = float((g_s - ref_s).norm() / (ref_s.norm() + 1e-30))
        e_n = abs(g_n - ref_n) / (ref_n + 1e-30)
        out[k] = max(e_s, e_n)
    return out


def reference_lowprec_errors(npz, lp, mode):
    """(logits rel-L2, {name: gradient digest error}) of the reference's OWN low-precision run ``mode`` (``amp_bf16`` /
    ``amp_fp16`` / ``pure_bf16`` / ``pure_fp16`` in a ``vivit_*_lowprec.npz`` fixture) against its fp32 run ``npz`` --
    measured exactly like ``grad_digest_errors`` measures the HIP path."""
    ref_logits = torch.from_numpy(npz["logits"]).double()
    e_out = float((torch.from_numpy(lp[f"{mode}:logits"]).double() - ref_logits).norm() / ref_logits.norm())
    pre = f"{mode}:gs:"
    stored = {k[len(pre):]: (lp[k], lp[f"{mode}:gn:" + k[len(pre):]]) for k in lp.files if k.startswith(pre)}
    return e_out, grad_digest_errors(npz, stored)


def reference_lowprec_yardstick(npz, lp, prec):
    """Like-for-like yardstick of the 16-bit protocol for element type ``prec`` ('bf16' / 'fp16'): per quantity the LARGER
    of the reference's two own low-precision deviations on the same inputs -- ``amp_<prec>`` (torch.autocast: matmuls
    in 16 bits, residual stream / LayerNorm / softmax in fp32) and ``pure_<prec>`` (module and clip cast to the type:
    everything, the residual stream included, stored in 16 bits -- which is how the HIP path stores it)."""
    a_out, a = reference_lowprec_errors(npz, lp, "amp_" + prec)
    p_out, p = reference_lowprec_errors(npz, lp, "pure_" + prec)
    return max(a_out, p_out), {k: max(a[k], p[k]) for k in a}


def assert_within_reference_lowprec(tag, e_out, errs, ref_out, ref_errs, factor=2.0, out_cap=None, floor=2e-4, grad_cap=None,
                                    flip_noise=False):
    """The parity protocol of SURVEY section 7 / BASELINE.md section 2 for the 16-bit kernels: "<= 2x the reference's own
    low-precision error on the same inputs", against ``reference_lowprec_yardstick`` (``ref_*``):
      * logits: under ``out_cap`` and within ``factor`` (2x) of the yardstick;
      * gradients as a population: the median deviation within ``factor`` of the yardsticks' median;
      * EVERY single gradient: within ``factor`` (2x) of ITS OWN like-for-like yardstick, and under the absolute
        ``grad_cap`` when one is given (a bound that does not move with the reference's noisier run: drift shows).
    Round 4 removed the floor that let a parameter be measured against the MEDIAN yardstick instead of its own: the one
    parameter that needed it, ``pos_embedding``, was a measurement artefact of the digest (256 evenly spaced entries of a
    tensor whose energy sits in 1 row of 197, see ``digest_idx``), not a property of the kernels.
    ``floor`` (2e-4) keeps quantities whose reference deviation is at fp32 round-off from dividing by ~0.
    ``flip_noise`` (the BatchNorm'd ReLU stacks of the CNN encoders only): there a gradient's deviation is a handful of
    DISCRETE events -- ReLU masks of activations that are 0 +- round-off flipping -- i.e. a Poisson draw per parameter, and a
    parameter for which the yardstick run happened to draw (almost) none is held to the typical (median) yardstick instead.
    Returns (worst (name, ratio) against its own yardstick, median ratio)."""
    assert e_out <= factor * ref_out + floor, (tag, "logits", e_out, ref_out)
    if out_cap is not None:
        assert e_out <= out_cap, (tag, "logits", e_out)
    med_ref = float(np.median(list(ref_errs.values())))
    med = float(np.median(list(errs.values())))
    assert med <= factor * med_ref + floor, (tag, "median gradient deviation", med, med_ref)
    worst = ("", 0.0)
    for k, e in errs.items():
        yard = max(ref_errs[k], med_ref) if flip_noise else ref_errs[k]
        if e / (yard + floor) > worst[1]:
            worst = (k, e / (yard + floor))
        assert e <= factor * yard + floor, (tag, k, e, ref_errs[k], med_ref)
        if grad_cap is not None:
            assert e <= grad_cap, (tag, k, "absolute cap", e, grad_cap)
    return worst, med / (med_ref + 1e-30)
