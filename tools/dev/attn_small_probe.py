"""Dev probe: short-sequence attention (dvt_attention_fwd / _bwd on the frametransformer encoder's shape) with and without
probability dropout, against the generic kernels (DVT_ATTN_SMALL=0 in a dev build)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dvt_amd
from dvt_amd import ops
from dvt_amd import functional as F

def t(f, n=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (S, H, N, dh) in ((2, 2, 14, 448), (2, 8, 15, 112), (8, 8, 32, 64), (2, 2, 14, 64)):
    inner = H * dh
    qkv = torch.randn(N, S, 3 * inner, device="cuda").to(torch.bfloat16)          # seq-first
    q, k, v = [qkv[:, :, i * inner:(i + 1) * inner].view(N, S, H, dh).permute(1, 2, 0, 3) for i in range(3)]
    o = torch.empty(N, S, H, dh, device="cuda", dtype=torch.bfloat16).permute(1, 2, 0, 3)
    do = torch.randn(N, S, H, dh, device="cuda").to(torch.bfloat16).permute(1, 2, 0, 3)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = [dqkv[:, :, i * inner:(i + 1) * inner].view(N, S, H, dh).permute(1, 2, 0, 3) for i in range(3)]
    for p in (0.0, 0.5):
        drop = None if p == 0 else (p, F._rng.tensor(q.device), F._rng.take(S * H * N * N))
        lse = ops.attention_fwd(q, k, v, o, dh ** -0.5, drop)
        a = t(lambda: ops.attention_fwd(q, k, v, o, dh ** -0.5, drop))
        b = t(lambda: ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, dh ** -0.5, drop))
        print(f"S={S} H={H} N={N} dh={dh} dropout={p}: fwd {a:.1f} us  bwd {b:.1f} us")
