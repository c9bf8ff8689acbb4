"""Dev probe: one training step of the reference-default FrameTransformer (R(2+1)D-18 encoder on 14 chunks of
12 x 112^2 frames per sample, frame_transformer.py:192-210) at the reference batch size 2 (config.yaml:2)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import dvt_amd
from dvt_amd.models.frame_transformer import FrameTransformer

torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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
    return loss

for _ in range(2):
    step()
torch.cuda.synchronize(); t = time.perf_counter()
n = 3
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print(f"FrameTransformer(vid) B={B}: {dt*1e3:.1f} ms/step, {B/dt:.2f} samples/s, loss {float(loss):.4f}, "
      f"peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
