"""Dev probe: dvt_conv3x3_c64 on the layer-1 shape (256 frames of 56^2) against the implicit GEMM; DVT_PROBE_LIB selects
another build of the library."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops, _lib as L
if os.environ.get("DVT_PROBE_LIB"):
    L.LIB_PATH = os.environ["DVT_PROBE_LIB"]
N, H, W = 256, 56, 56
x = torch.randn(N * H * W, 64, device="cuda").to(torch.bfloat16)
w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
wp = ops.conv_weight_pack(w, 576, torch.bfloat16)

def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = t(lambda: ops.conv3x3_c64(x, wp, N, H, W))
b = t(lambda: ops.conv2d_implicit(x, wp, N, 64, H, W, 64, 3, 1, 1))
print(f"{os.environ.get('DVT_PROBE_LIB', 'product')}: halo {a:.1f} us, implicit GEMM {b:.1f} us")
c = t(lambda: ops.conv3x3_c64(x, wp, N, H, W, want_stats=True))
d = t(lambda: ops.conv2d_implicit(x, wp, N, 64, H, W, 64, 3, 1, 1, want_stats=True))
print(f"with BatchNorm partial sums: halo {c:.1f} us, implicit GEMM {d:.1f} us")
