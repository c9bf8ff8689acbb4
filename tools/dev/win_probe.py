"""Dev: the LDS-window / halo kernels of R(2+1)D-18's layer 1 at the frametransformer shape (28 clips of 12 x 56^2 = 336 frames),
a few launches each on rotating operands -- the program the SQ counter passes of tools/dev/win_pmc.sh run over
(profiles/r06_conv3x1_bound.md).  usage: win_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dvt_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, T, H, W = 28, 12, 56, 56
Lp = H * W
rows = N * T * Lp
dt = torch.bfloat16
x144 = [torch.randn(rows, 144, device="cuda").to(dt) for _ in range(2)]
x64 = [torch.randn(rows, 64, device="cuda").to(dt) for _ in range(2)]
mean, istd = torch.zeros(144, device="cuda"), torch.ones(144, device="cuda")
aff = (mean, istd, torch.ones(144, device="cuda"), torch.zeros(144, device="cuda"), 0, True)
w_t = torch.randn(64, 144, 3, 1, device="cuda") * 0.05               # temporal 144 -> 64
wp_t = ops.conv_weight_pack(w_t, ops.conv2d_implicit_k(144, 64, (3, 1)), dt)
wd_t = ops.conv_weight_pack_dgrad(w_t, dt)
w_s = torch.randn(144, 64, 3, 3, device="cuda") * 0.04               # spatial 64 -> 144
wp_s = ops.conv_weight_pack(w_s, ops.conv2d_implicit_k(64, 144, (3, 3)), dt)
wd_s = ops.conv_weight_pack_dgrad(w_s, dt)
w_c = torch.randn(64, 64, 3, 3, device="cuda") * 0.04
wp_c = ops.conv_weight_pack(w_c, ops.conv2d_implicit_k(64, 64, (3, 3)), dt)
dw_t = torch.empty(64, 144, 3, 1, device="cuda")
dw_s = torch.empty(144, 64, 3, 3, device="cuda")
NF = N * T


def one(i):
    a, b = x144[i & 1], x64[i & 1]
    ops.conv3x1_fwd(a, wp_t, N, T, Lp, want_stats=True, affine=aff)                      # conv3x1_fwd_kernel
    p = ops.conv3x1_wgrad(a, b, N, T, Lp, dw_t, defer_reduce=True, affine=aff); p.valid = 0   # conv3x1_wgrad_kernel
    ops.conv3x1_stream_bn_bwd(b, wd_t, a, aff, N, T, Lp, True)                           # conv3x3_stream<64,144,3,1/2>
    ops.conv3x3_stream(b, wp_s, NF, H, W, 64, 144, want_stats=True)                      # conv3x3_stream<64,144,9>
    ops.conv3x3_stream(a, wd_s, NF, H, W, 144, 64)                                       # conv3x3_stream<144,64,9>
    p = ops.conv3x3_c64_wgrad(b, a, NF, H, W, dw_s, defer_reduce=True, Cout=144); p.valid = 0  # conv3x3_c64_wgrad<.., 4 / 5>
    ops.conv3x3_c64(b[:256 * Lp], wp_c, 256, H, W)                                                  # ResNet-18 layer 1 (256 frames)


for i in range(2):
    one(i)
torch.cuda.synchronize()
evs = []
for i in range(reps):
    one(i)
torch.cuda.synchronize()
print("done", reps)
