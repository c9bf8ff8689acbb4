"""Dev probe: do HBM-bound kernels (LayerNorm bwd, attention bwd) hide under MFMA-bound weight-gradient GEMMs
when issued on a second stream?  Prints sequential vs concurrent time for one layer's backward mix."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops

torch.manual_seed(0)
M, d = 50432, 512
dev = "cuda"
x = torch.randn(M, d, device=dev).bfloat16()
dy = torch.randn(M, d, device=dev).bfloat16()
dy4 = torch.randn(M, 4 * d, device=dev).bfloat16()
x4 = torch.randn(M, 4 * d, device=dev).bfloat16()
g = torch.ones(d, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, g, torch.zeros(d, device=dev))
qkv = torch.randn(256, 197, 3, 8, 64, device=dev).bfloat16()
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
o_mem = torch.empty(256, 197, 8, 64, device=dev, dtype=torch.bfloat16)
o = o_mem.permute(0, 2, 1, 3)
lse = ops.attention_fwd(q, k, v, o, 0.125)
do = torch.randn_like(o_mem).permute(0, 2, 1, 3)
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
w_ff1 = torch.empty(4 * d, d, device=dev)
w_ff2 = torch.empty(d, 4 * d, device=dev)

def wgrads():
    ops.linear_wgrad(dy4, x, out=w_ff1)          # FF1 wgrad  [2048 x 512], K = 50432
    ops.linear_wgrad(dy, x4, out=w_ff2)          # FF2 wgrad

def hbm_mix():
    ops.layernorm_bwd(dy, x, g, mean, rstd, dx_add=dy)
    ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, 0.125)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dx_add=dy)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

side = torch.cuda.Stream()
ws_main = ops._ws_cache if hasattr(ops, "_ws_cache") else None
def concurrent():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hbm_mix()                                # uses no workspace shared with the GEMMs except layernorm/attention scratch
    wgrads()
    torch.cuda.current_stream().wait_stream(side)

a, b = timeit(wgrads), timeit(hbm_mix)
c = timeit(concurrent)
print(f"wgrads {a:.3f} ms  hbm mix {b:.3f} ms  sum {a+b:.3f} ms  concurrent {c:.3f} ms")
