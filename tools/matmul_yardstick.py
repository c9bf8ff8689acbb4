"""Dev-only yardstick (NOT product code): vendor GEMM (torch.matmul -> hipBLASLt) on the
metric shapes, to know what this hardware sustains on them."""
import torch, time
M = 50432
shapes = [("qkv_fwd", M, 1536, 512), ("ff1_fwd", M, 2048, 512), ("ff2_fwd", M, 512, 2048), ("proj_fwd", M, 512, 512),
          ("square", 4096, 4096, 4096)]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"vendor {name:10s} {us:8.1f} us {2.0*m*n*k/us/1e6:8.1f} TF/s")
# pure write / copy bandwidth yardsticks
x = torch.empty(M * 2048, device="cuda", dtype=torch.bfloat16)
y = torch.empty_like(x)
for fn, label, bytes_ in ((lambda: x.zero_(), "memset 206MB", x.numel() * 2), (lambda: y.copy_(x), "copy 206MB (r+w)", 2 * x.numel() * 2)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{label:20s} {us:8.1f} us {bytes_/us/1e6:6.2f} TB/s")
