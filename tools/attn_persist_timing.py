"""Dev probe: per-wave stamps of one iteration of the persistent dk/dv kernel (tools/attn_timing.sh's library)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd  # noqa: F401
from dvt_amd import _lib as L, ops
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libdvt_hip_timing.so")
import numpy as np
S, H, N, dh = 256, 8, 197, 64
dt = torch.bfloat16
qkv = torch.randn(S, N, 3, H, dh, device="cuda").to(dt)
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
o = torch.empty(S, N, H, dh, device="cuda", dtype=dt).permute(0, 2, 1, 3)
do = torch.randn(S, N, H, dh, device="cuda").to(dt).permute(0, 2, 1, 3)
lse = ops.attention_fwd(q, k, v, o, dh ** -0.5)
for _ in range(3):
    ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, dh ** -0.5)
torch.cuda.synchronize()
off = ((S * H * N + 1) & ~1) * 4
ws = ops.workspace(off + 256 * 128 * 8, q.device)
t = ws[off: off + 256 * 128 * 8].view(torch.int64).view(256, 128).cpu().numpy().astype(np.float64)
W = 13
for wg in (0, 100, 200):
    it = t[wg, : W * 5].reshape(W, 5)
    base = it[:, 0].min()
    print(f"workgroup {wg}, iteration 3: per wave [arrive at top, after barrier, after stores+issue, after compute, after tail] ticks from first arrival")
    for w in range(W):
        print(f"   wave {w:2d}: " + " ".join(f"{x - base:7.0f}" for x in it[w]))
