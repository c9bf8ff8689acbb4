import sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops
N, T, H, W = 2, 12, 8, 7
Lp = H * W
dtype = torch.bfloat16
g = torch.Generator().manual_seed(1)
z = torch.randn(N * T * Lp, 144, generator=g).to(dtype).cuda()
w = (torch.randn(64, 144, 3, 1, generator=g) * (2.0 / (144 * 3)) ** 0.5).cuda()
wp = ops.conv_weight_pack(w, ops.conv2d_implicit_k(144, 64, (3, 1)), dtype)
y = ops.conv3x1_fwd(z, wp, N, T, Lp).float()
zr = z.float().view(N, T, Lp, 144).permute(0, 3, 1, 2)
ref = torch.nn.functional.conv2d(zr, w.to(dtype).float(), None, 1, (1, 0)).permute(0, 2, 3, 1).reshape(-1, 64)
err = (y - ref).abs()
bad = err > 0.05
print("bad elements", int(bad.sum()), "of", bad.numel())
rows = bad.any(1).nonzero().flatten().tolist()
print("bad rows", len(rows), rows[:60])
for r in rows[:12]:
    n, rem = divmod(r, T * Lp); t, p = divmod(rem, Lp)
    print("row", r, "clip", n, "frame", t, "pixel", p, "seg", p // 8, "sx", p % 8, "bad cols", bad[r].nonzero().flatten().tolist()[:20])
# per-tap contributions: which tap is missing?
for kt in range(3):
    wk = w.to(dtype).float().clone(); 
    for j in range(3):
        if j != kt: wk[:, :, j] = 0
    part = torch.nn.functional.conv2d(zr, wk, None, 1, (1, 0)).permute(0, 2, 3, 1).reshape(-1, 64)
    r0 = rows[0] if rows else 0
    print("tap", kt, "contribution at first bad row (col of first bad):", float(part[r0][bad[r0].nonzero()[0]]) if rows else None)
if rows:
    r0 = rows[0]; c0 = int(bad[r0].nonzero()[0])
    print("y", float(y[r0, c0]), "ref", float(ref[r0, c0]), "diff", float(y[r0, c0] - ref[r0, c0]))
print("---- tile 0 (clip 0, seg 0): bad positions / channels")
S = 8
badpos = {}
for t in range(T):
    for sx in range(S):
        r = t * Lp + sx
        cols = bad[r].nonzero().flatten().tolist()
        if cols:
            badpos[t * S + sx] = cols
print("bad positions:", sorted(badpos))
import collections
cnt = collections.Counter()
for pos, cols in badpos.items():
    for c in cols:
        cnt[c % 4] += 1
print("bad channel index mod 4 histogram:", dict(cnt))
cntu = collections.Counter()
for pos, cols in badpos.items():
    for c in cols:
        cntu[c // 16] += 1
print("by co block:", dict(cntu))
# is the bad value equal to the ref value of another channel / position?
r0 = 2 * Lp + 0
print("y row  :", [round(float(v), 3) for v in y[r0][:16]])
print("ref row:", [round(float(v), 3) for v in ref[r0][:16]])
