"""Dev probe (GPU box): per-parameter ratio of the HIP bf16 error to the reference's own autocast-bf16 error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from tests.util import golden, rel_l2, fill_state_from_numpy, digest_inputs, grad_digest_errors, reference_lowprec_errors
import dvt_amd
from dvt_amd import functional as F
from dvt_amd.models.vit import ViViT

def vivit(tag, mode, scale=1.0):
    g = golden(f"vivit_{tag}_digest.npz"); lp = golden(f"vivit_{tag}_lowprec.npz")
    cfg, x, y = digest_inputs(g)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[mode]
    net = ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["frames"], dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"], compute_dtype=dt)
    fill_state_from_numpy(net.named_parameters(), int(g["fill_seed"]))
    net = net.cuda()
    logits = net(x.cuda()); loss = F.bce_with_logits(logits, y.cuda()); loss.backward(torch.tensor(scale, device="cuda"))
    grads = {k: p.grad / scale for k, p in net.named_parameters()}
    errs = grad_digest_errors(g, grads)
    for ym in ("amp_" + mode, "pure_bf16"):
        if f"{ym}:logits" not in lp.files: continue
        ro, re = reference_lowprec_errors(g, lp, ym)
        r = sorted(((errs[k] / re[k], k, errs[k], re[k]) for k in errs), reverse=True)
        print(f"== {tag}/{mode} scale {scale} vs {ym}: logits ours {rel_l2(logits, torch.from_numpy(g['logits'])):.2e} ref {ro:.2e}; median ref {np.median(list(re.values())):.2e} median ours {np.median(list(errs.values())):.2e}")
        for t in r[:6]: print("   %.2f %s ours %.2e ref %.2e" % t)

vivit("c2", "bf16"); vivit("c2", "fp16", 256.0); vivit("c2", "fp16", 8192.0); vivit("metric", "bf16"); vivit("metric", "fp16", 8192.0)

import tests.test_gpu_frame_transformer as T
for mode in ("sum", "distil", "post_sum"):
    net = T._make_ft(mode, torch.bfloat16)
    g = torch.Generator().manual_seed(4)
    vid = torch.randn(2, 4, 2, 3, 16, 16, generator=g); img = torch.randn(2, 4, 3, 32, 32, generator=g)
    target = (torch.rand(2, 19, generator=g) < 0.3).float()
    P = T._oracle_params(net)
    rl, ro = T._oracle_ft_loss(P, mode, img, vid, target); rl.backward()
    Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ll, lo = T._oracle_ft_loss(Q, mode, img, vid, target)
    ll.float().backward()
    loss = net.training_step((target.cuda(), img.cuda(), vid.cuda()), 0); loss.backward()
    named = dict(net.named_parameters())
    rows = []
    for k in P:
        if k in named and P[k].grad is not None and float(P[k].grad.abs().max()) > 0:
            rows.append((rel_l2(named[k].grad, P[k].grad) / (rel_l2(Q[k].grad, P[k].grad) + 1e-30), k, rel_l2(named[k].grad, P[k].grad), rel_l2(Q[k].grad, P[k].grad)))
    rows.sort(reverse=True)
    print(f"== ft/{mode}: median ours {np.median([r[2] for r in rows]):.2e} median oracle-amp {np.median([r[3] for r in rows]):.2e}")
    for t in rows[:10]: print("   %.2f %s ours %.2e ref %.2e" % t)
