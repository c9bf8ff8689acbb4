"""Dev probe: per-shape convolution / BatchNorm time of one FrameTransformer(vid) training step (HIP-event brackets, ops.set_profiler)."""
import sys, torch
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dvt_amd
from dvt_amd import ops
from dvt_amd.models.frame_transformer import FrameTransformer
from bench import EventProfiler

torch.manual_seed(0)
B = 2
net = FrameTransformer(batch_size=B, seq_len=13, cls=1, model="vid", opt="adamW", learning_rate=5e-6, weight_decay=0.09,
                       momentum=0.005).cuda().train()
vid = torch.randn(B, 13, 12, 3, 112, 112, device="cuda")
target = (torch.rand(B, 19, device="cuda") < 0.3).float()
opt = net.configure_optimizers()

def step():
    opt.zero_grad(set_to_none=True)
    loss = net.training_step((target, None, vid), 0)
    loss.backward()
    opt.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
prof = EventProfiler()
prof.overhead_ms = 0.0
ops.set_profiler(prof)
step()
torch.cuda.synchronize()
ops.set_profiler(None)
rows = [(ms, k, n) for k, (ms, fl, n) in prof.summary().items() if k[0] in ("conv", "hbm")]
rows.sort(reverse=True)
tot = 0.0
for ms, k, n in rows[:120]:
    fl = prof.summary()[k][1]
    frac = (fl / (ms * 1e-3) / 1e12 / 2500.0) if k[0] == "conv" else 0.0
    tot += ms
    print(f"{ms*1e3/n:9.1f} us x{n:3d}  total {ms*1e3:8.1f} us  frac {frac:5.3f}  {k}")
print("sum ms", tot)
