"""Dev probe: upper bound of running the step as two independent half-batch chains on two streams inside one hipGraph
(two model replicas of B = 4 each, each with its own optimizer step) against one chain of B = 8."""
import sys, torch
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dvt_amd
from dvt_amd.dp import FlatParameters
from dvt_amd.models.vit import ViViT
from dvt_amd import functional as F
from dvt_amd.graph import capture_step


def make(B, seed):
    torch.manual_seed(seed)
    net = ViViT(224, 16, 19, 32, dim=512, depth=4, heads=8, dim_head=64, compute_dtype=torch.bfloat16).cuda().train()
    flat = FlatParameters(net, compute_dtype=torch.bfloat16)
    flat.sync_compute_copy()
    x = torch.randn(B, 32, 3, 224, 224).to(torch.bfloat16).cuda()
    y = (torch.rand(B, 19) < 0.2).float().cuda()
    gloss = torch.ones((), device="cuda")

    def step():
        flat.zero_grad()
        loss = F.bce_with_logits(net(x), y)
        loss.backward(gloss)
        flat.finish_backward()
        flat.adamw_step(lr=5e-6, weight_decay=0.09)
        return loss
    return step


def timeit(replay, n=100):
    for _ in range(10):
        replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


one = make(8, 1)
r1, _ = capture_step(one, warmup=2)
t1 = timeit(r1)
a, b = make(4, 2), make(4, 3)
s2 = torch.cuda.Stream()


def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        lb = b()
    la = a()
    torch.cuda.current_stream().wait_stream(s2)
    return la


def serial():
    b()
    return a()


r2, _ = capture_step(both, warmup=2)
t2 = timeit(r2)
r3, _ = capture_step(serial, warmup=2)
t3 = timeit(r3)
print(f"one chain B=8: {t1:.3f} ms;  two B=4 chains on two streams: {t2:.3f} ms;  two B=4 chains on one stream: {t3:.3f} ms")
