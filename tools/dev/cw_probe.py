"""Dev: dvt_conv3x3_c64_wgrad at the pyramid shape (256 frames of 56^2), timed.  The split quoted in conv3x3_wgrad.hip
(DMA stream alone / LDS reads + MFMAs alone) came from a build with two debug switches in the kernel (DVT_CW_DBG = 1: no
fragment reads / MFMAs, 2: no tile DMA after the first), not kept in the product source."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dvt_amd import ops, _lib as L
if os.environ.get("DVT_LIB"):
    L.LIB_PATH = os.environ["DVT_LIB"]
N, H, W = int(os.environ.get("CW_N", "256")), 56, 56          # CW_N=336 CW_COUT=144: the frametransformer's layer 1
Cout = int(os.environ.get("CW_COUT", "64"))
x = torch.randn(N * H * W, 64, device="cuda").to(torch.bfloat16)
dz = torch.randn(N * H * W, Cout, device="cuda").to(torch.bfloat16)
dw = torch.empty(Cout, 64, 3, 3, device="cuda")
_f = ops.conv3x3_c64_wgrad
ops.conv3x3_c64_wgrad = lambda *a, **k: _f(*a, Cout=Cout, **k)
for _ in range(3):
    pend = ops.conv3x3_c64_wgrad(x, dz, N, H, W, dw, defer_reduce=True); pend.valid = 0
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        pend = ops.conv3x3_c64_wgrad(x, dz, N, H, W, dw, defer_reduce=True); pend.valid = 0
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
ts.sort()
print(f"lib={os.environ.get('DVT_LIB', 'in-tree')} N={N} Cout={Cout} DVT_CW_DBG={os.environ.get('DVT_CW_DBG', '0')}: {ts[2]:.1f} us per call (all groups, reduces of all but the last)")
