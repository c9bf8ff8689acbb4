"""Timing of dvt_conv3x3_stream against the implicit GEMM at the R(2+1)D-18 layer-1 shapes (336 frames of 56^2)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa: F401  (puts dvt_amd on the path)
from dvt_amd import ops

def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

N, H, W = 336, 56, 56
for Cin, Cout in ((64, 144), (144, 64)):
    x = torch.randn(N * H * W, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05)
    wp = ops.conv_weight_pack(w, 9 * Cin, torch.bfloat16)
    fl = 2.0 * N * H * W * Cout * 9 * Cin
    us = t(lambda: ops.conv3x3_stream(x, wp, N, H, W, Cin, Cout, want_stats=Cout == 144))
    ui = t(lambda: ops.conv2d_implicit(x, wp, N, Cin, H, W, Cout, 3, 1, 1, want_stats=Cout == 144))
    print(f"{Cin}->{Cout}: stream {us:.1f} us ({fl / us / 1e6:.0f} TF/s)   implicit {ui:.1f} us ({fl / ui / 1e6:.0f} TF/s)")
