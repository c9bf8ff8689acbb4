"""Dev probe: would a grouped weight-gradient launch pay?  FF1 + FF2 (and QKV + proj) weight gradients of one layer,
(a) back to back with the planner's split, (b) on two streams with half the slices each (half the GPU per GEMM: half the
slab bytes and reduce work in total) -- an upper-bound emulation of one grouped launch."""
import sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops
torch.manual_seed(0)
M = 50432
def mk(N, K):
    return (torch.randn(M, N, device="cuda").bfloat16(), torch.randn(M, K, device="cuda").bfloat16(),
            torch.empty(N, K, device="cuda"))
pairs = {"ff1+ff2": (mk(2048, 512), mk(512, 2048)), "qkv+proj": (mk(1536, 512), mk(512, 512))}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def wg(t, split=0):
    dy, x, out = t
    Mr, N = dy.shape; K = x.shape[1]
    ops.gemm(dy, x, N, K, Mr, a_kmajor=False, b_kmajor=False, lda=dy.stride(0), ldb=x.stride(0), out=out,
             out_dtype=torch.float32, split_k=split)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (a, b) in pairs.items():
    seq = timeit(lambda: (wg(a), wg(b)))
    for sa, sb in ((8, 8), (10, 10), (6, 12), (12, 6)):
        def par():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1): wg(a, sa)
            with torch.cuda.stream(s2): wg(b, sb)
            cur.wait_stream(s1); cur.wait_stream(s2)
        print(f"{name}: sequential {seq:.1f} us; two streams split {sa}/{sb}: {timeit(par):.1f} us")
