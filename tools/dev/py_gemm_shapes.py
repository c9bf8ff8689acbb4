"""Dev probe: per-shape GEMM / convolution time of one pyramid-workload training step (HIP-event brackets)."""
import sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd import ops
from dvt_amd.dp import FlatParameters
from dvt_amd.models.pyramid_vivit import PyramidViViT
from dvt_amd import functional as F
from bench import EventProfiler

torch.manual_seed(0)
B = 8
net = PyramidViViT(224, 19, 32, dim=512, depth=4, heads=8, dim_head=64, audio_tokens=0, audio_dim=128, distill=False,
                   compute_dtype=torch.bfloat16).cuda().train()
flat = FlatParameters(net, compute_dtype=torch.bfloat16)
flat.sync_compute_copy()
x = torch.randn(B, 32, 3, 224, 224).to(torch.bfloat16).cuda()
y = (torch.rand(B, 19) < 0.2).float().cuda()

def step():
    flat.zero_grad()
    loss = F.bce_with_logits(net(x), y)
    loss.backward()
    flat.finish_backward()

for _ in range(2):
    step()
torch.cuda.synchronize()
prof = EventProfiler()
prof.overhead_ms = 0.0
ops.set_profiler(prof)
step()
torch.cuda.synchronize()
ops.set_profiler(None)
rows = [(ms, k, n, fl) for k, (ms, fl, n) in prof.summary().items() if k[0] == "gemm"]
rows.sort(reverse=True)
for ms, k, n, fl in rows[:45]:
    print(f"{ms*1e3/n:9.1f} us x{n:3d}  {fl/ms/1e9:7.1f} TF/s  {k}")
