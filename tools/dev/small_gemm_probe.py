"""Dev probe: the launch-bound GEMMs of the temporal encoder (264 rows) in isolation: per-launch time from a chain of
launches in one graph, with warm and with cold weights.  (The per-workgroup phase stamps quoted in DESIGN 4.1 came from a
-DDVT_SMALL_TIMING build of gemm_small.hip that lived in the round-3 commits before this one; they are not in the product source.)"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dvt_amd  # noqa: F401
from dvt_amd import _lib as L, ops
lib_t = os.path.join(ROOT, "tools", "_bin", "libdvt_hip_stiming.so")
timing = os.path.exists(lib_t) and "--no-stamps" not in sys.argv
if timing:
    L.LIB_PATH = lib_t
import numpy as np

dt = torch.bfloat16
M = 264
shapes = [("QKV fwd", 1536, 512, "none"), ("proj+res", 512, 512, "res"), ("FF1+GELU", 2048, 512, "gelu"), ("FF2+res", 512, 2048, "res")]
for name, N, K, kind in shapes:
    x = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    bias = torch.randn(N, device="cuda") if kind != "none" else None
    res = torch.randn(M, N, device="cuda").to(dt)
    aux = torch.empty(M, N, device="cuda", dtype=dt)
    _lin = ops.linear_fwd

    def call(x=x, w=w, bias=bias, res=res, aux=aux, kind=kind):
        if kind == "res":
            return _lin(x, w, bias, epilogue=L.EPI_RESIDUAL, residual=res)
        if kind == "gelu":
            return _lin(x, w, bias, epilogue=L.EPI_GELU, aux=aux)
        return _lin(x, w, bias)

    class _O:
        pass
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            y = call()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    line = f"{name:9s} M={M} N={N} K={K}: {e0.elapsed_time(e1) * 1e3 / 50:.2f} us per launch (50 in one graph)"
    # cold weights: every launch reads a different weight matrix from a pool larger than the Infinity Cache (the situation
    # inside a training step, where a layer's weights were last touched a whole step ago)
    pool = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(dt) for _ in range(max(8, int(600e6 / (N * K * 2))))]
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for i in range(200):
            y = call(w=pool[i % len(pool)])
    g2.replay(); torch.cuda.synchronize()
    e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
    line += f"; cold weights ({len(pool)} matrices): {e0.elapsed_time(e1) * 1e3 / 200:.2f} us"
    del pool
    if timing:
        lib = L.load()
        lib.dvt_debug_small_timing_buffer.argtypes = [ctypes.c_void_p]
        nb = 4096
        tb = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
        lib.dvt_debug_small_timing_buffer(tb.data_ptr())
        call(); torch.cuda.synchronize()
        lib.dvt_debug_small_timing_buffer(0)
        t = tb.view(nb, 8).cpu().numpy().astype(np.float64)
        t = t[t[:, 4] > 0]
        base = t[:, 0].min()
        ph = [np.median(t[:, i + 1] - t[:, i]) for i in range(4)]
        line += (f"; {len(t)} workgroups; ticks (median): issue {ph[0]:.0f}, wait chunk 0 {ph[1]:.0f}, main loop {ph[2]:.0f}, "
                 f"epilogue {ph[3]:.0f}; first start -> last end {t[:, 4].max() - base:.0f}; start spread {np.percentile(t[:, 0] - base, 90):.0f}")
    print(line)
