"""Dev: dvt_conv3x1_fwd at the frametransformer shape (28 clips of 12 x 56^2, virtual BatchNorm), timed.  The split quoted in
DESIGN 4.5 came from a build with run-time switches in the kernel (DVT_TF_DBG bits: 1 no window transform, 2 no fragment
reads / MFMAs, 4 no output staging / stores / statistics, 8 no window requests after the first two), not kept in the product."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dvt_amd import ops, _lib as L
if os.environ.get('DVT_LIB'):
    L.LIB_PATH = os.environ['DVT_LIB']
N, T, Lp = 28, 12, 56 * 56
xs = [torch.randn(N * T * Lp, 144, device="cuda").to(torch.bfloat16) for _ in range(3)]   # rotate: no Infinity-Cache re-reads
wp = (torch.randn(64, 432, device="cuda") / 20).to(torch.bfloat16)
mean, var = torch.zeros(144, device="cuda"), torch.ones(144, device="cuda")
aff = (mean, var.rsqrt(), torch.ones(144, device="cuda"), torch.zeros(144, device="cuda"), 144, True)
use_aff = os.environ.get("TF_AFFINE", "1") == "1"
def run(i):
    return ops.conv3x1_fwd(xs[i % 3], wp, N, T, Lp, want_stats=True, affine=aff if use_aff else None)
for i in range(3):
    run(i)
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(9):
        run(i)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 9 * 1e3)
ts.sort()
print(f"lib={os.environ.get('DVT_LIB', 'in-tree')} DVT_TF_DBG={os.environ.get('DVT_TF_DBG', '0')} affine={int(use_aff)}: fwd {ts[2]:.1f} us per launch")
dz = torch.randn(N * T * Lp, 64, device="cuda").to(torch.bfloat16)
dw = torch.empty(64, 144, 3, 1, device="cuda")
def runw(i):
    pend = ops.conv3x1_wgrad(xs[i % 3], dz, N, T, Lp, dw, defer_reduce=True, affine=aff if use_aff else None); pend.valid = 0
for i in range(3):
    runw(i)
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(9):
        runw(i)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 9 * 1e3)
ts.sort()
print(f"   wgrad {ts[2]:.1f} us per launch")
