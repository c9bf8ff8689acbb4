"""Dev probe: per-workgroup timeline of the N = 197 forward attention kernel (needs tools/attn_timing.sh's library):
stage (K / V into LDS, incl. the barrier) / first tile computed / first tile stored / end, per-CU residency."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dvt_amd  # noqa: F401
from dvt_amd import _lib as L, ops
L.LIB_PATH = os.path.join(ROOT, "tools", "_bin", "libdvt_hip_timing.so")
import numpy as np

S, H, N, dh = 256, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 197, 64
dt = torch.bfloat16
qkv = torch.randn(S, N, 3, H, dh, device="cuda").to(dt)
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
o = torch.empty(S, N, H, dh, device="cuda", dtype=dt).permute(0, 2, 1, 3)
for _ in range(3):
    ops.attention_fwd(q, k, v, o, dh ** -0.5)
tb = torch.zeros(S * H * 8, dtype=torch.int64, device="cuda")
lib = L.load()
lib.dvt_debug_attn_timing_buffer.argtypes = [ctypes.c_void_p]
assert lib.dvt_debug_attn_timing_buffer(tb.data_ptr()) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.attention_fwd(q, k, v, o, dh ** -0.5)
e1.record()
torch.cuda.synchronize()
t = tb.view(S * H, 8).cpu().numpy().astype(np.float64)
print(f"instrumented launch: {e0.elapsed_time(e1) * 1e3:.1f} us; kernel span {(t[:, 4].max() - t[:, 0].min()):.0f} ticks "
      f"(tick = shader clock)")
print("per-workgroup phase lengths in ticks, median / p10 / p90 over", S * H, "workgroups")
for a, b_, nm in ((0, 1, "stage: K / V -> LDS, barrier"), (1, 2, "wave 0: first 16-query tile computed"),
                  (2, 3, "wave 0: first tile stored"), (3, 4, "wave 0: second tile (compute + store)"), (0, 4, "whole workgroup (wave 0)")):
    d = t[:, b_] - t[:, a]
    print(f"  {nm:44s} {np.median(d):9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 90):9.0f}")
hw, xcc = t[:, 6].astype(np.int64), t[:, 7].astype(np.int64) & 0xF
cu, se, sh = (hw >> 8) & 0xF, (hw >> 13) & 0x7, (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 50 + cu
conc = []
for kx in np.unique(key):
    m = key == kx
    st, en = t[m, 0], t[m, 4]
    span = en.max() - st.min()
    ev = sorted([(a, 1) for a in st] + [(b_, -1) for b_ in en])
    cur = mx = 0
    for _, dlt in ev:
        cur += dlt; mx = max(mx, cur)
    conc.append(((en - st).sum() / span, m.sum(), span, mx))
conc = np.array(conc)
print(f"{len(conc)} distinct CUs; workgroups per CU {conc[:,1].min():.0f}..{conc[:,1].max():.0f}; resident workgroups per CU: "
      f"time-average median {np.median(conc[:,0]):.2f} (p10 {np.percentile(conc[:,0],10):.2f}, p90 {np.percentile(conc[:,0],90):.2f}), "
      f"maximum {conc[:,3].max():.0f}; per-CU busy span median {np.median(conc[:,2]):.0f} ticks, max {conc[:,2].max():.0f}")
kx = np.unique(key)[len(np.unique(key)) // 2]
m = key == kx
base = t[m, 0].min()
print("one CU's workgroups (start, staged, first tile computed, end), ticks from its first start:")
for i in np.argsort(t[m, 0]):
    row = t[m][i]
    print(f"   start {row[0]-base:8.0f}  staged {row[1]-base:8.0f}  tile {row[2]-base:8.0f}  end {row[4]-base:8.0f}")
