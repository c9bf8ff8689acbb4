"""Dev probe: which Python lines still launch ATen kernels / device copies inside one training step of a workload?
torch.profiler (CPU activities, with_stack) around ONE eager step; prints the aten ops that touch the device
(copy_, contiguous, clone, zero_, fill_, add, ...) aggregated by their innermost repo frame.

    python tools/dev/aten_trace.py pyramid|vivit|crossmodal|frametransformer
"""
import collections
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "pyramid"
args = types.SimpleNamespace(batch=8, dtype="bf16", grad_dtype="fp32", bucket_mb=32.0)
W = bench.build_workload(args, wl, 0, None)
step = W["step"]
for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
shapes = {}
WATCH = ("aten::copy_", "aten::clone", "aten::zero_", "aten::fill_", "aten::add", "aten::add_", "aten::mul", "aten::cat",
         "aten::_to_copy", "aten::sum", "aten::div", "aten::sub")
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    frame = "?"
    for fr in (ev.stack or []):
        if ROOT in fr and "aten_trace" not in fr:
            frame = fr.replace(ROOT + "/", "")
            break
    key = (ev.name, frame)
    agg[key] += 1
    shapes.setdefault(key, str(ev.input_shapes)[:80])
for (name, frame), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:16s} {frame}   {shapes[(name, frame)]}")
print("total watched aten calls per step:", sum(agg.values()))
