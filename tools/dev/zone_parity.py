"""Dev probe (GPU box): every ViViT golden config in bf16 / fp16 with the fp32 zone (temporal stack + heads) on and off:
logits deviation and each gradient's ratio to ITS OWN like-for-like yardstick (tests/util.reference_lowprec_yardstick)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.util import golden, rel_l2, fill_state_from_numpy, digest_inputs, grad_digest_errors, reference_lowprec_yardstick
from dvt_amd import functional as F
from dvt_amd.models.vit import ViViT


def run(tag, prec, zone):
    g = golden(f"vivit_{tag}_digest.npz"); lp = golden(f"vivit_{tag}_lowprec.npz")
    cfg, x, y = digest_inputs(g)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[prec]
    kw = dict(activation_checkpointing=True) if tag == "longclip" else {}
    net = ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["frames"], dim=cfg["dim"], depth=cfg["depth"],
                heads=cfg["heads"], dim_head=cfg["dim_head"], compute_dtype=dt, **kw)
    net.zone_f32 = zone
    fill_state_from_numpy(net.named_parameters(), int(g["fill_seed"]))
    net = net.cuda()
    scale = 1.0 if prec == "bf16" else 1024.0
    logits = net(x.cuda())
    F.bce_with_logits(logits, y.cuda()).backward(torch.tensor(scale, device="cuda"))
    F.ln_flush(True)
    errs = grad_digest_errors(g, {k: p.grad / scale for k, p in net.named_parameters()})
    yo, yard = reference_lowprec_yardstick(g, lp, prec)
    e_out = rel_l2(logits, torch.from_numpy(g["logits"]))
    r = sorted(((errs[k] / (yard[k] + 1e-30), k, errs[k], yard[k]) for k in errs), reverse=True)
    med = np.median(list(errs.values())) / np.median(list(yard.values()))
    print(f"{tag:9s} {prec} zone_f32={int(zone)}: logits {e_out:.2e} (yardstick {yo:.2e}); median ratio {med:.2f}; worst own-yardstick ratios: "
          + ", ".join(f"{a:.2f} {k.replace('_transformer', '')[-38:]}" for a, k, _, _ in r[:4]))


if "--fold" in sys.argv:                              # the folded single-query form of the last space layer at any size
    F.CLS_FOLD_MIN_ROWS = 0
    print("CLS fold forced")
for tag in ("c2", "metric", "longclip"):
    for prec in ("bf16", "fp16"):
        if tag == "longclip" and prec == "bf16":
            continue
        try:
            for zone in (False, True):
                run(tag, prec, zone)
        except Exception as e:
            print(tag, prec, "failed:", type(e).__name__, str(e)[:300])
