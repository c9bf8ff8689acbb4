"""Dev probe: weight-gradient GEMM shapes of the metric workload (mn-major x mn-major, split-K), TF/s."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops
torch.manual_seed(0)
M = 50432
for name, N, K in (("qkv", 1536, 512), ("ff1", 2048, 512), ("ff2", 512, 2048), ("proj", 512, 512)):
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.empty(N, K, device="cuda")
    f = lambda: ops.linear_wgrad(dy, x, out=out)
    f()
    ref = dy[:4096].float().t() @ x[:4096].float()
    chk = ops.linear_wgrad(dy[:4096].contiguous(), x[:4096].contiguous())
    err = float((chk - ref).norm() / ref.norm())
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"CFG={os.environ.get('DVT_GEMM_CFG','-')} wgrad {name:5s} {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s  rel err {err:.1e}")
