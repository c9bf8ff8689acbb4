"""Dev probe: who launches the `__amd_rocclr_copyBuffer` kernels inside a step?  Every function of dvt_amd.ops is wrapped
in a torch.profiler.record_function range, one eager step runs under torch.profiler (CPU + device activities), and every
runtime call whose name mentions Memcpy / Memset (plus every copyBuffer / fillBuffer kernel) is attributed to the
innermost enclosing ops.* range by CPU time.

    python tools/dev/copy_trace.py pyramid|vivit|frametransformer
"""
import collections
import functools
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from torch.profiler import profile, ProfilerActivity, record_function
import dvt_amd
from dvt_amd import ops, functional

def wrap(mod, prefix):
    for name in dir(mod):
        f = getattr(mod, name)
        if isinstance(f, types.FunctionType) and f.__module__ == mod.__name__ and not name.startswith("_"):
            def mk(f, label):
                @functools.wraps(f)
                def g(*a, **k):
                    with record_function(label):
                        return f(*a, **k)
                return g
            setattr(mod, name, mk(f, f"{prefix}.{name}"))

wrap(ops, "ops")
for name in dir(functional):                     # autograd Functions: forward / backward ranges
    cls = getattr(functional, name)
    if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        for meth in ("forward", "backward"):
            f = cls.__dict__.get(meth)
            if isinstance(f, staticmethod):
                def mk(f, label):
                    def g(*a, **k):
                        with record_function(label):
                            return f(*a, **k)
                    return staticmethod(g)
                setattr(cls, meth, mk(f.__func__, f"F.{name}.{meth}"))

wl = sys.argv[1] if len(sys.argv) > 1 else "pyramid"
args = types.SimpleNamespace(batch=8, dtype="bf16", grad_dtype="fp32", bucket_mb=32.0)
W = bench.build_workload(args, wl, 0, None)
step = W["step"]
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
evs = list(prof.events())
ranges = [(e.time_range.start, e.time_range.end, e.name) for e in evs if e.name.startswith(("ops.", "F."))]
agg = collections.Counter()
names = collections.Counter()
for e in evs:
    n = e.name
    if not any(t in n for t in ("emcpy", "emset", "copyBuffer", "fillBuffer")):
        continue
    names[n] += 1
    if "Buffer" in n:                    # device-side record: attributed through its runtime call instead
        continue
    t = e.time_range.start
    best = None
    for s, en, rn in ranges:
        if s <= t <= en and (best is None or en - s < best[0]):
            best = (en - s, rn)
    outer = None
    for s, en, rn in ranges:
        if s <= t <= en and rn.startswith("F.") and (outer is None or en - s < outer[0]):
            outer = (en - s, rn)
    agg[(n, best[1] if best else "<outside ops>", outer[1] if outer else "-")] += 1
print("events by name:", dict(names))
for (n, r, o), c in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {n:28s} {r:38s} {o}")
