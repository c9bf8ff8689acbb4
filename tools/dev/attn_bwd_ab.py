"""Dev: launch time of the attention backward at the metric shape (256 frames x 8 heads x 197 tokens) for the one-pass kernel
and for the dq + dk/dv pair; interleaved repetitions, medians."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dvt_amd import ops
B, H, N, dh = 256, 8, 197, 64
qkv = (torch.randn(B, N, 3, H, dh) * 0.7).to(torch.bfloat16).cuda()
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
o = torch.empty(B, N, H, dh, dtype=torch.bfloat16, device="cuda").permute(0, 2, 1, 3)
lse = ops.attention_fwd(q, k, v, o, dh ** -0.5)
do = torch.randn(B, N, H, dh).to(torch.bfloat16).cuda().permute(0, 2, 1, 3)
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
variants = {"pair": dict(two_pass=True), "one-pass": dict()}
res = {k_: [] for k_ in variants}
for rep in range(7):
    for name, kw in variants.items():
        for _ in range(2):
            ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, dh ** -0.5, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, dh ** -0.5, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
for name, v_ in res.items():
    v_.sort()
    print(f"{name:24s} median {v_[len(v_) // 2]:7.1f} us   min {v_[0]:7.1f}")
