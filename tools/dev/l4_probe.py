"""Dev: the implicit-GEMM convolutions of R(2+1)D-18's layers 3 - 4 at the frametransformer shape (28 clips), one at a time,
for a kernel trace: which launches (grid, kernel) a layer takes and how long each is.  usage: l4_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dvt_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dt = torch.bfloat16
# (frames N, H, W, Cin, Cout, kernel, pad): the view every (2+1)D half is run in (temporal: N = clips, H = T, W = pixels)
SHAPES = [
    ("l4 spatial fwd 512->1152", 56, 7, 7, 512, 1152, (3, 3), (1, 1)),
    ("l4 spatial dgrad 1152->512", 56, 7, 7, 1152, 512, (3, 3), (1, 1)),
    ("l4 temporal fwd 1152->512", 28, 2, 49, 1152, 512, (3, 1), (1, 0)),
    ("l4 temporal dgrad 512->1152", 28, 2, 49, 512, 1152, (3, 1), (1, 0)),
    ("l3 spatial fwd 256->576", 84, 14, 14, 256, 576, (3, 3), (1, 1)),
    ("l3 spatial dgrad 576->256", 84, 14, 14, 576, 256, (3, 3), (1, 1)),
    ("l3 temporal fwd 576->256", 28, 3, 196, 576, 256, (3, 1), (1, 0)),
    ("l3 temporal dgrad 256->576", 28, 3, 196, 256, 576, (3, 1), (1, 0)),
]
ev = []
for name, N, H, W, Ci, Co, k, pad in SHAPES:
    x = [torch.randn(N * H * W, Ci, device="cuda").to(dt) for _ in range(2)]
    w = torch.randn(Co, Ci, *k, device="cuda") * 0.03
    wp = ops.conv_weight_pack(w, ops.conv2d_implicit_k(Ci, Co, k), dt)
    for i in range(2):
        ops.conv2d_implicit(x[i & 1], wp, N, Ci, H, W, Co, k, 1, pad)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        ops.conv2d_implicit(x[i & 1], wp, N, Ci, H, W, Co, k, 1, pad)
    b.record()
    torch.cuda.synchronize()
    fl = 2.0 * N * H * W * Co * Ci * k[0] * k[1]
    us = a.elapsed_time(b) * 1e3 / reps
    print(f"{name:32s} rows {N*H*W:6d}  {us:7.1f} us/launch-group  {fl/us/1e6/2500:.3f} of the MFMA peak", flush=True)
