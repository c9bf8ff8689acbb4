import sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops
N, T, H, W = 2, 12, 8, 7
Lp = H * W
dtype = torch.bfloat16
g = torch.Generator().manual_seed(1)
z0 = torch.randn(N * T * Lp, 144, generator=g).to(dtype).cuda()
w = (torch.randn(64, 144, 3, 1, generator=g) * (2.0 / (144 * 3)) ** 0.5).cuda()
wp = ops.conv_weight_pack(w, ops.conv2d_implicit_k(144, 64, (3, 1)), dtype)
def run(tag, z, wsel=None):
    ww = w if wsel is None else wsel
    wpp = ops.conv_weight_pack(ww, ops.conv2d_implicit_k(144, 64, (3, 1)), dtype)
    y = ops.conv3x1_fwd(z, wpp, N, T, Lp).float()
    zr = z.float().view(N, T, Lp, 144).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(zr, ww.to(dtype).float(), None, 1, (1, 0)).permute(0, 2, 3, 1).reshape(-1, 64)
    bad = (y - ref).abs() > 0.05
    print(tag, "bad", int(bad.sum()), "rel", float((y - ref).norm() / ref.norm()))
z = z0.clone(); z[:, 128:] = 0; run("channels 0..127 only", z)
z = z0.clone(); z[:, :128] = 0; run("channels 128..143 only", z)
for kk in range(4):
    z = torch.zeros_like(z0); z[:, kk * 32:(kk + 1) * 32] = z0[:, kk * 32:(kk + 1) * 32]; run(f"channels {kk*32}..{kk*32+31} only", z)
for kt in range(3):
    ws = torch.zeros_like(w); ws[:, :, kt] = w[:, :, kt]; run(f"tap {kt} only", z0, ws)
