"""Times dvt_attn_cls_fwd / _bwd and the head-wise products at the metric shape (S = 256 frames x 197 rows x 512, 8 heads).
    python3 tools/dev/attn_cls_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dvt_amd  # noqa: E402
from dvt_amd import ops  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S, N, d, H = 256, 197, 512, 8
g = torch.Generator().manual_seed(0)
x = torch.randn(S, N, d, generator=g).bfloat16().cuda()
gam, bet = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (0.1 * torch.randn(d, generator=g)).cuda()
R = (torch.randn(S, H, d, generator=g) / d ** 0.5).cuda()
dM = torch.randn(S, H, d, generator=g).cuda()
W = (torch.randn(3 * 512, d, generator=g) / d ** 0.5).bfloat16().cuda()
q = torch.randn(S, 512, generator=g).bfloat16().cuda()
buf = torch.zeros(3 * 512, d, device="cuda")


def timed(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:16s} {1e3 * e0.elapsed_time(e1) / it:8.1f} us")


A, lse, P, mean, rstd = ops.attn_cls_fwd(x, gam, bet, 1e-5, R)
timed("attn_cls_fwd", lambda: ops.attn_cls_fwd(x, gam, bet, 1e-5, R))
timed("attn_cls_bwd", lambda: ops.attn_cls_bwd(x, gam, bet, 1e-5, R, A, lse, P, mean, rstd, dM))
timed("heads_expand", lambda: ops.heads_expand(q, W[512:1024], H, 0.125))
timed("heads_contract", lambda: ops.heads_contract(A, W[1024:], 1.0, gam, bet))
timed("heads_outer", lambda: ops.heads_outer(q, A, buf[512:1024], 1.0, gam, bet))
