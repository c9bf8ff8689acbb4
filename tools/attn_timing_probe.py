"""Dev probe: per-workgroup phase stamps of the dk/dv attention kernel (needs tools/attn_timing.sh's library)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd  # noqa: F401
from dvt_amd import _lib as L, ops
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libdvt_hip_timing.so")
import numpy as np

S, H, N, dh = 256, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 197, 64
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
ws = ops.workspace(off + S * H * 64, q.device)
t = ws[off: off + S * H * 64].view(torch.int64).view(S * H, 8).cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
names = ["start", "staged", "tile0 begin", "tile0 end", "tile1 begin", "tile1 end"]
print("per-workgroup phase lengths in shader-clock ticks (s_memtime), median / p10 / p90 over", S * H, "workgroups")
for a, b_, nm in ((0, 1, "stage (loads + stats + barrier)"), (2, 3, "key tile 0 (7 steps)"), (4, 5, "key tile 1 (7 steps)"), (0, 5, "whole workgroup")):
    d = t[:, b_] - t[:, a]
    print(f"  {nm:36s} {np.median(d):9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 90):9.0f}")
hw, xcc = t[:, 6].astype(np.int64), t[:, 7].astype(np.int64) & 0xF
cu, se = (hw >> 8) & 0xF, (hw >> 13) & 0x7          # HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
key = xcc * 1000 + se * 100 + ((hw >> 12) & 1) * 50 + cu
conc = []
for kx in np.unique(key):
    m = key == kx
    st, en = t[m, 0], t[m, 5]
    span = en.max() - st.min()
    conc.append(((en - st).sum() / span, m.sum(), span))
conc = np.array(conc)
print(f"{len(conc)} distinct CUs; workgroups per CU {conc[:,1].min():.0f}..{conc[:,1].max():.0f}; time-averaged resident workgroups "
      f"per CU: median {np.median(conc[:,0]):.2f} (p10 {np.percentile(conc[:,0],10):.2f}, p90 {np.percentile(conc[:,0],90):.2f}); "
      f"per-CU busy span median {np.median(conc[:,2]):.0f} ticks, max {conc[:,2].max():.0f}")
kx = np.unique(key)[len(np.unique(key)) // 2]
m = key == kx
order = np.argsort(t[m, 0])
base = t[m, 0].min()
print("one CU's workgroups (start, staged, end of last tile), ticks from its first start; simd/wave slot from HW_ID:")
for i in order:
    row = t[m][i]
    print(f"   start {row[0]-base:8.0f}  staged {row[1]-base:8.0f}  end {row[5]-base:8.0f}   wave_id {int(row[6]) & 0xF} simd {(int(row[6]) >> 4) & 3}")
print("per shader engine (XCC, SE): CUs, max / time-averaged resident workgroups")
sekey = xcc * 10 + se
for kx in np.unique(sekey)[:6]:
    m = sekey == kx
    ev = sorted([(a, 1) for a in t[m, 0]] + [(b_, -1) for b_ in t[m, 5]])
    cur = mx = 0
    for _, dlt in ev:
        cur += dlt
        mx = max(mx, cur)
    span = t[m, 5].max() - t[m, 0].min()
    print(f"   xcc {kx // 10} se {kx % 10}: {len(np.unique(key[m]))} CUs, {m.sum()} workgroups, max {mx}, avg {(t[m,5]-t[m,0]).sum()/span:.1f}")
