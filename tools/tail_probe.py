"""Dev probe: does the partial last round of workgroups cost a full round?  FF2-forward shape at M = 50432 (394 tiles
of 256x256 on 256 CUs = 1.54 rounds) against M = 65536 (2.0 rounds) and M = 32768 (1.0 round)."""
import sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops, _lib as L
for M in (32768, 50432, 65536):
    for N, K in ((512, 2048), (512, 512), (1536, 512), (2048, 512)):
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        f = lambda: ops.linear_fwd(x, w)
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"M={M} N={N} K={K}: tiles {tiles} rounds {tiles/256:.2f}  {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")
