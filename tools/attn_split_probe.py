"""Dev probe: long-sequence attention (N = 325: K/V images of 90 KiB allow one workgroup per CU) -- does running the
kernel on two key halves (45 KiB each, three workgroups per CU) beat the single launch?  Kernel time only."""
import sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops
S, H, N, dh = 512, 8, 325, 64
qkv = torch.randn(S, N, 3, H, dh, device="cuda").half()
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
o = torch.empty(S, N, H, dh, device="cuda", dtype=torch.float16).permute(0, 2, 1, 3)
o2 = torch.empty(S, N, H, dh, device="cuda", dtype=torch.float16).permute(0, 2, 1, 3)
def timeit(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
full = timeit(lambda: ops.attention_fwd(q, k, v, o, 0.125))
h = 176
def halves():
    ops.attention_fwd(q, k[:, :, :h], v[:, :, :h], o, 0.125)
    ops.attention_fwd(q, k[:, :, h:], v[:, :, h:], o2, 0.125)
half = timeit(halves)
def thirds():
    for a in (0, 112, 224):
        ops.attention_fwd(q, k[:, :, a:a + 112], v[:, :, a:a + 112], o, 0.125)
third = timeit(thirds)
print(f"N=325 fwd: single launch {full:.1f} us; two key halves {half:.1f} us; three thirds {third:.1f} us (+ combine)")
