"""Dev probe: per-workgroup phase lengths (s_memtime ticks) of the LDS-DMA GEMM on the metric's forward shapes."""
import os, sys, ctypes, torch
sys.path.insert(0, "/root/repo")
import dvt_amd  # noqa: F401
from dvt_amd import ops, _lib as L
L.LIB_PATH = os.environ.get("DVT_PROBE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libdvt_hip_gtiming.so")
import numpy as np
lib = L.load()
tb = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
lib.dvt_debug_gemm_timing_buffer.argtypes = [ctypes.c_void_p]
assert lib.dvt_debug_gemm_timing_buffer(tb.data_ptr()) == 0
M = 50432
for name, N, K, epi in [("qkv", 1536, 512, L.EPI_NONE), ("ff1+gelu", 2048, 512, L.EPI_GELU), ("ff2+res", 512, 2048, L.EPI_RESIDUAL),
                        ("proj+res", 512, 512, L.EPI_RESIDUAL), ("square", 4096, 4096, L.EPI_NONE)]:
    m = 4096 if name == "square" else M
    x = torch.randn(m, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(N, device="cuda")
    res = torch.randn(m, N, device="cuda").bfloat16() if epi == L.EPI_RESIDUAL else None
    aux = torch.empty(m, N, device="cuda", dtype=torch.bfloat16) if epi == L.EPI_GELU else None
    for _ in range(3):
        ops.linear_fwd(x, w, b, epilogue=epi, residual=res, aux=aux)
    torch.cuda.synchronize()
    tiles = ((m + 255) // 256) * (N // 256)
    t = tb[: tiles * 4].view(tiles, 4).cpu().numpy().astype(np.float64)
    d = np.diff(t, axis=1)
    med = np.median(d, axis=0)
    wt = tb[(1 << 19): (1 << 19) + tiles * 2].view(tiles, 2).cpu().numpy().astype(np.float64)
    wm = np.median(wt, axis=0) / (K // 64)
    print(f"{name:10s} per k-tile (wave 0): DMA wait {wm[0]:6.0f}, barrier wait {wm[1]:6.0f} ticks")
    print(f"{name:10s} {tiles:5d} tiles, {K // 64:3d} k-tiles: prologue issue {med[0]:7.0f}  main loop {med[1]:8.0f} ({med[1] / (K // 64):6.0f}/k-tile)  "
          f"epilogue {med[2]:7.0f}  whole {np.median(t[:, 3] - t[:, 0]):8.0f} ticks")
