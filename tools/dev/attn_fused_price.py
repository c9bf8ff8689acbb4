"""Dev probe: what would fusing the QKV projection into the attention forward save?
Its only saving over the two launches is that q, k, v are not READ BACK from HBM by the attention kernel (they must still
be written for the backward).  Measured here: the space-encoder attention forward of the metric shape (256 frames x 8 heads,
N = 197, dh = 64) (a) streaming its packed qkv from HBM (three rotating 155 MB buffers: more than the 256 MiB Infinity
Cache between two uses), (b) on 64-frame quarters whose 39 MB of qkv stay on die (same buffer every launch), x 4.
The difference is the upper bound of the saving; tools/dev/gemm_storewave_price.hip prices the other side -- the QKV
projection on the small tiles a (frame, head) workgroup would have to use."""
import sys
import torch
sys.path.insert(0, "/root/repo")
import dvt_amd  # noqa: F401,E402
from dvt_amd import ops  # noqa: E402

H, N, dh = 8, 197, 64
dt = torch.bfloat16
torch.manual_seed(0)


def views(qkv):
    return tuple(qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))


def timeit(f, n=30):
    for i in range(4):
        f(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        f(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


S = 256
bufs = [torch.randn(S, N, 3, H, dh, device="cuda").to(dt) for _ in range(3)]
outs = [torch.empty(S, N, H, dh, device="cuda", dtype=dt).permute(0, 2, 1, 3) for _ in range(3)]
full = timeit(lambda i: ops.attention_fwd(*views(bufs[i % 3]), outs[i % 3], dh ** -0.5))
same = timeit(lambda i: ops.attention_fwd(*views(bufs[0]), outs[0], dh ** -0.5))
res = {}
for Sq in (32, 64, 128):
    q = torch.randn(Sq, N, 3, H, dh, device="cuda").to(dt)
    o = torch.empty(Sq, N, H, dh, device="cuda", dtype=dt).permute(0, 2, 1, 3)
    res[Sq] = timeit(lambda i: ops.attention_fwd(*views(q), o, dh ** -0.5), 60)
u = S * H * N * dh * 2
print(f"attention forward, 256 frames x 8 heads, N = {N}:")
print(f"  qkv streamed from HBM (3 rotating buffers)   {full:6.1f} us  ({4 * u / full / 1e6:.2f} TB/s algorithmic)")
print(f"  same 155 MB buffer every launch              {same:6.1f} us")
for Sq, t in res.items():
    print(f"  {Sq:3d}-frame launches on one resident buffer ({Sq * N * 3 * H * dh * 2 / 1e6:5.1f} MB of qkv): {t:6.1f} us each = {t * S / Sq:6.1f} us per 256 frames")
