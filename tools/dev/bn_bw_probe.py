"""Dev: BatchNorm backward passes at equal bytes and different widths (is the 144-plane case slow, or is every 300 MB map?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dvt_amd import ops
for rows, C in ((1053696, 144), (2370816, 64), (1053696, 64), (1185408, 128), (526848, 288), (263424, 576)):
    z = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
    mean, invstd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for _ in range(3):
        ops.bn_bwd(dy, z, None, mean, invstd, g, True, True, False, beta=b)
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.bn_bwd(dy, z, None, mean, invstd, g, True, True, False, beta=b)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    ts.sort()
    mb = rows * C * 2 / 1e6
    print(f"rows {rows:8d} C {C:4d}: {mb:6.0f} MB per map, bn_bwd (stats + finalize + apply) {ts[2]:7.1f} us -> {5 * mb / ts[2]:.2f} TB/s over 5 streams")
    del z, dy
