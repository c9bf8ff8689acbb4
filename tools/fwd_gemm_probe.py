"""Dev probe: forward-Linear GEMM shapes of the metric workload, TF/s (env DVT_GEMM_BDIRECT=0/1 selects the variant)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops, _lib as L
torch.manual_seed(0)
M = 50432
shapes = [("qkv", 1536, 512, L.EPI_NONE), ("ff1+gelu", 2048, 512, L.EPI_GELU), ("ff2+res", 512, 2048, L.EPI_RESIDUAL),
          ("proj+res", 512, 512, L.EPI_RESIDUAL), ("square", 4096, 4096, L.EPI_NONE)]
for name, N, K, epi in shapes:
    m = 4096 if name == "square" else M
    x = torch.randn(m, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(N, device="cuda")
    res = torch.randn(m, N, device="cuda").bfloat16() if epi == L.EPI_RESIDUAL else None
    aux = torch.empty(m, N, device="cuda", dtype=torch.bfloat16) if epi == L.EPI_GELU else None
    f = lambda: ops.linear_fwd(x, w, b, epilogue=epi, residual=res, aux=aux)
    y = f()
    ref = x.float() @ w.float().t()
    if epi == L.EPI_RESIDUAL: ref = ref + res.float()
    if epi == L.EPI_GELU: ref = torch.nn.functional.gelu(ref)
    err = float((y.float() - ref).norm() / ref.norm())
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"BDIRECT={os.environ.get('DVT_GEMM_BDIRECT','0')} {name:10s} {us:7.1f} us  {2*m*N*K/us/1e6:7.1f} TF/s  rel err {err:.1e}")
