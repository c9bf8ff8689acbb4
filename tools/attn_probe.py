"""Dev probe: the space-encoder attention launches of the metric shape in isolation (256 frames x 8 heads, N = 197,
dh = 64, packed qkv as the QKV GEMM writes it).  Prints per-launch times (forward; backward = dq + dk/dv launches);
run under `rocprofv3 --kernel-trace --pmc ...` for the SQ counters of exactly these kernels."""
import sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd  # noqa: F401
from dvt_amd import ops, _lib as L
import os
if os.environ.get('DVT_PROBE_LIB'):
    L.LIB_PATH = os.environ['DVT_PROBE_LIB']

S, H, N, dh = 256, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 197, 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = torch.bfloat16
torch.manual_seed(0)
qkv = torch.randn(S, N, 3, H, dh, device="cuda").to(dt)
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
o = torch.empty(S, N, H, dh, device="cuda", dtype=dt).permute(0, 2, 1, 3)
do = torch.randn(S, N, H, dh, device="cuda").to(dt).permute(0, 2, 1, 3)


def timeit(f, n):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


lse = ops.attention_fwd(q, k, v, o, dh ** -0.5)
fwd = timeit(lambda: ops.attention_fwd(q, k, v, o, dh ** -0.5), iters)
bwd = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, dh ** -0.5), iters)
u = S * H * N * dh * 2
print(f"N={N}: fwd {fwd:.1f} us ({4 * u / fwd / 1e6:.2f} TB/s algorithmic), bwd (dq + dk/dv) {bwd:.1f} us "
      f"({(6 * u + 7 * u) / bwd / 1e6:.2f} TB/s)")

if os.environ.get('DVT_PROBE_NOPARITY'):
    sys.exit(0)
# parity of the big launch against fp32 autograd on a few frames (first, middle, last)
ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, dh ** -0.5)
torch.cuda.synchronize()
for f in sorted({0, 1, S // 2, S - 2, S - 1}):
    qf, kf, vf = (t[f].float().detach().requires_grad_(True) for t in (q, k, v))
    ref = torch.softmax(qf @ kf.transpose(-1, -2) * dh ** -0.5, -1) @ vf
    ref.backward(do[f].float())
    errs = [(a[f].float() - b).abs().max().item() / (b.abs().max().item() + 1e-9)
            for a, b in ((o, ref), (dq, qf.grad), (dk, kf.grad), (dv, vf.grad))]
    print(f"frame {f}: rel max err o {errs[0]:.1e} dq {errs[1]:.1e} dk {errs[2]:.1e} dv {errs[3]:.1e}", "OK" if max(errs) < 2e-2 else "MISMATCH")
