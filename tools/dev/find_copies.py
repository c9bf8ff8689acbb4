"""Dev: where do the device-to-device copies of a training step come from?  (torch.profiler with stacks)"""
import sys, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd.dp import FlatParameters
from dvt_amd.models.vit import ViViT
from dvt_amd import functional as F
from torch.profiler import profile, ProfilerActivity

net = ViViT(224, 16, 19, 32, dim=512, depth=4, heads=8, dim_head=64, compute_dtype=torch.bfloat16).cuda().train()
flat = FlatParameters(net, compute_dtype=torch.bfloat16)
flat.sync_compute_copy()
x = torch.randn(8, 32, 3, 224, 224).to(torch.bfloat16).cuda()
y = (torch.rand(8, 19) < 0.2).float().cuda()
g = torch.ones((), device="cuda")

def step():
    flat.zero_grad()
    loss = F.bce_with_logits(net(x), y)
    loss.backward(g)
    flat.finish_backward()
    flat.adamw_step(lr=1e-5, weight_decay=0.1)

for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = list(prof.events())
for e in evs:
    if "emcpy" in e.name or "copyBuffer" in e.name or e.name in ("aten::copy_", "aten::clone", "aten::_to_copy"):
        print(e.name, e.input_shapes, "|", [s for s in (e.stack or [])][:5])
print("---- counts")
names = {}
for e in evs:
    names[e.name] = names.get(e.name, 0) + 1
for k, v in sorted(names.items(), key=lambda kv: -kv[1]):
    if "emcpy" in k or "copy" in k.lower() or "fill" in k.lower() or "zero" in k.lower():
        print(v, k[:120])
