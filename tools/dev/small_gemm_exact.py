"""Dev probe: gemm_small.hip against float64 on bf16-exact operands with fp32 output (no output rounding):
the only error left is fp32 accumulation order (~1e-6)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dvt_amd
from dvt_amd import ops, _lib as L
g = torch.Generator().manual_seed(1)
for (M, N, K) in [(264, 1536, 512), (264, 512, 2048), (264, 2048, 512), (8, 512, 512), (264, 512, 512)]:
    x = torch.randn(M, K, generator=g).bfloat16(); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    wt = (torch.randn(K, N, generator=g) / math.sqrt(K)).bfloat16(); dy = torch.randn(M, N, generator=g).bfloat16()
    xd, wd, wtd, dyd = x.cuda(), w.cuda(), wt.cuda(), dy.cuda()
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    y = ops.gemm(xd, wd, M, N, K, a_kmajor=True, b_kmajor=True, lda=K, ldb=K, out_dtype=torch.float32)
    e1 = rel(y, x.double() @ w.double().t())
    dx = ops.gemm(xd, wtd, M, N, K, a_kmajor=True, b_kmajor=False, lda=K, ldb=N, out_dtype=torch.float32)
    e2 = rel(dx, x.double() @ wt.double())
    dw = ops.linear_wgrad(dyd, xd)
    e3 = rel(dw, dy.double().t() @ x.double())
    print(f"M{M} N{N} K{K}: fwd {e1:.2e} dgrad {e2:.2e} wgrad {e3:.2e}")
