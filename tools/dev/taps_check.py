"""Dev: a small conv+BN+ReLU layer (Cin 16 -> 24, 3x3) through the general-tap implicit path and the explicit path: outputs and
gradients of the two against each other and against the fp32 CPU reference."""
import sys, os, torch
import torch.nn.functional as TF
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa: F401
from dvt_amd import functional as F

def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())

dtype = torch.bfloat16
g = torch.Generator().manual_seed(31)
N, Cin, Cout, H, W = 3, 16, 24, 10, 12
x = torch.randn(N, Cin, H, W, generator=g)
conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False)
bn = torch.nn.BatchNorm2d(Cout)
with torch.no_grad():
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (Cin * 9)) ** 0.5)
    bn.weight.copy_(1 + 0.1 * torch.randn(Cout, generator=g)); bn.bias.copy_(0.1 * torch.randn(Cout, generator=g))
res = torch.randn(N, Cout, H, W, generator=g)
gy = torch.randn(N, Cout, H, W, generator=g)
xr = x.to(dtype).float().clone().requires_grad_(True)
wr = conv.weight.detach().to(dtype).float().clone().requires_grad_(True)
rr = res.to(dtype).float().clone().requires_grad_(True)
z = TF.conv2d(xr, wr, None, 1, 1)
ref = torch.relu(TF.batch_norm(z, torch.zeros(Cout), torch.ones(Cout), bn.weight.detach().clone(), bn.bias.detach().clone(), True, 0.1, 1e-5) + rr)
ref.backward(gy.to(dtype).float())
outs = {}
from dvt_amd import ops
rec = {}
_bn_bwd, _ldg, _c2i = ops.bn_bwd, ops.linear_dgrad, ops.col2im
def bn_bwd(*a, **k):
    r = _bn_bwd(*a, **k)
    rec.setdefault(cur[0], {})["dz"] = r[0].clone(); rec[cur[0]]["mean"] = a[3].clone(); rec[cur[0]]["invstd"] = a[4].clone(); rec[cur[0]]["z"] = a[1].clone()
    return r
def ldg(*a, **k):
    r = _ldg(*a, **k)
    rec[cur[0]]["dcol"] = r.clone()
    return r
ops.bn_bwd, ops.linear_dgrad = bn_bwd, ldg
cur = [None]
for taps in (True, False):
    cur[0] = taps
    F.GENERAL_TAPS = taps
    c2, b2 = torch.nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False), torch.nn.BatchNorm2d(Cout)
    c2.load_state_dict(conv.state_dict()); b2.load_state_dict(bn.state_dict())
    c2, b2 = c2.cuda(), b2.cuda().train()
    xd = x.to(dtype).permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda().requires_grad_(True)
    rd = res.to(dtype).permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda().requires_grad_(True)
    y = F.conv_bn_act(xd, c2, b2, (N, Cin, H, W, False), relu=True, residual=rd, dtype=dtype)
    y.backward(gy.to(dtype).permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda())
    outs[taps] = (y.detach(), c2.weight.grad.clone(), xd.grad.clone())
    rn = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    print(f"general taps {taps}: y {rel(y, rn):.4f}  dW {rel(c2.weight.grad, wr.grad):.4f}  dx {rel(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin)):.4f}")
a, b = outs[True], outs[False]
print(f"implicit vs explicit: y {rel(a[0], b[0]):.5f}  dW {rel(a[1], b[1]):.5f}  dx {rel(a[2], b[2]):.5f}")
print("mask flips between the two:", int(((a[0] > 0) != (b[0] > 0)).sum()), "of", a[0].numel())

for key in ("z", "mean", "invstd", "dz", "dcol"):
    a_, b_ = rec[True][key], rec[False][key]
    if a_.shape != b_.shape:
        n = min(a_.shape[1], b_.shape[1]); a_, b_ = a_[:, :144], b_[:, :144]
    print(key, tuple(rec[True][key].shape), tuple(rec[False][key].shape), rel(a_, b_))
