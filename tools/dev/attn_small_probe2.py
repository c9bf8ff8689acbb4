"""Dev probe: GPU time of the short-sequence attention forward inside a captured graph (20 launches per replay)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dvt_amd
from dvt_amd import ops
S, H, N, dh = 2, 2, 14, 448
inner = H * dh
qkv = torch.randn(N, S, 3 * inner, device="cuda").to(torch.bfloat16)
q, k, v = [qkv[:, :, i * inner:(i + 1) * inner].view(N, S, H, dh).permute(1, 2, 0, 3) for i in range(3)]
o = torch.empty(N, S, H, dh, device="cuda", dtype=torch.bfloat16).permute(1, 2, 0, 3)
for _ in range(3): ops.attention_fwd(q, k, v, o, dh ** -0.5, None)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): ops.attention_fwd(q, k, v, o, dh ** -0.5, None)
for _ in range(3): g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("DVT_ATTN_SMALL_STOP", "full"), f"{e0.elapsed_time(e1) / 200 * 1e3:.2f} us per launch")
