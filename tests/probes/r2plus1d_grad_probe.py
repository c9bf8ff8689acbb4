"""Dev probe (GPU): where does the fp32 gradient of the R(2+1)D-18 stack leave the conv3d oracle?  Prints, per residual
block, the rel-L2 error of the block's OUTPUT gradient (HIP path vs oracle.cnn_path arithmetic with retain_grad)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as TF
import dvt_amd  # noqa
from oracle import cnn_path as C
from dvt_amd.models import video_resnet as VR
from tests.util import rel_l2


def main():
    T, HW, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 32, 2
    net = VR.r2plus1d_18(False, compute_dtype=torch.float32)
    rng = np.random.default_rng(91)
    with torch.no_grad():
        for name, p in net.named_parameters():
            a = rng.standard_normal(tuple(p.shape)).astype(np.float32)
            if p.dim() == 5:
                a *= np.float32(np.sqrt(2.0 / (p.shape[1] * p.shape[2] * p.shape[3] * p.shape[4])))
            elif p.dim() == 2:
                a *= np.float32(0.02)
            elif name.endswith("weight"):
                a = 1 + np.float32(0.1) * a
            else:
                a = np.float32(0.1) * a
            p.copy_(torch.from_numpy(a))
    x = torch.from_numpy(rng.standard_normal((N, 3, T, HW, HW)).astype(np.float32))
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k in P:
        if P[k].dtype.is_floating_point and "running" not in k:
            P[k].requires_grad_(True)
    # ---- oracle with retained block-output gradients
    kept = {}
    h = TF.conv3d(x, P["stem.0.weight"], None, (1, 2, 2), (0, 3, 3))
    h = torch.relu(C._bn3(h, P, "stem.1.", True))
    h = TF.conv3d(h, P["stem.3.weight"], None, 1, (1, 0, 0))
    h = torch.relu(C._bn3(h, P, "stem.4.", True))
    h.retain_grad(); kept["stem"] = h
    for li in range(4):
        for b in range(2):
            pre = f"layer{li + 1}.{b}."
            stride = 2 if (li > 0 and b == 0) else 1
            out = C._conv2plus1d(h, P, pre + "conv1.0.", stride, True)
            out = torch.relu(C._bn3(out, P, pre + "conv1.1.", True))
            out.retain_grad(); kept[pre + "mid"] = out
            out = C._conv2plus1d(out, P, pre + "conv2.0.", 1, True)
            out = C._bn3(out, P, pre + "conv2.1.", True)
            res = h
            if pre + "downsample.0.weight" in P:
                res = TF.conv3d(h, P[pre + "downsample.0.weight"], None, stride)
                res = C._bn3(res, P, pre + "downsample.1.", True)
            h = torch.relu(out + res)
            h.retain_grad(); kept[pre + "out"] = h
    ref = h.mean(dim=(2, 3, 4))
    gy = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * gy).sum().backward()
    # ---- device, hooks on the same tensors
    got, fwd = {}, {}
    orig = VR.BasicBlock.forward_ndhwc
    names = {}
    for li, layer in enumerate((net.layer1, net.layer2, net.layer3, net.layer4)):
        for b, blk in enumerate(layer):
            names[id(blk)] = f"layer{li + 1}.{b}."

    def patched(self, fm, dtype):
        pre = names[id(self)]
        y, N_, T_, H_, W_ = fm
        c1 = self.conv1[0]
        o = VR._spatial(fm, c1[0], c1[1], True, dtype)
        o = VR._temporal(o, c1[3], self.conv1[1], True, dtype)
        fwd[pre + "mid"] = (o[0].detach(), o[1:])
        o[0].register_hook(lambda g, k=pre + "mid": got.__setitem__(k, g.detach().clone()))
        residual = y
        if self.downsample is not None:
            ds, dbn = self.downsample[0], self.downsample[1]
            st = ds.stride
            Cc = ds.in_channels
            r = VR.F.subsample_nhwc(y, N_ * T_, Cc, H_, W_, (st[1], st[2]))
            Hs, Ws = (H_ - 1) // st[1] + 1, (W_ - 1) // st[2] + 1
            r = VR.F.subsample_nhwc(r, N_, Cc, T_, Hs * Ws, (st[0], 1))
            Ts = (T_ - 1) // st[0] + 1
            residual = VR.F.conv_bn_act_raw(r, ds.weight, dbn, (N_ * Ts, Cc, Hs, Ws, False), 1, 1, 0, relu=False, dtype=dtype)
        c2 = self.conv2[0]
        o = VR._spatial(o, c2[0], c2[1], True, dtype)
        o = VR._temporal(o, c2[3], self.conv2[1], True, dtype, residual=residual)
        fwd[pre + "out"] = (o[0].detach(), o[1:])
        o[0].register_hook(lambda g, k=pre + "out": got.__setitem__(k, g.detach().clone()))
        return o

    VR.BasicBlock.forward_ndhwc = patched
    net = net.cuda().train()
    out = net.features(x.cuda())
    print("features rel", rel_l2(out, ref))
    out.backward(gy.cuda())

    def to_ndhwc(t):
        return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1])

    for k in kept:
        if k in got:
            print(f"{k:18s} fwd rel {rel_l2(fwd[k][0], to_ndhwc(kept[k].detach())):.2e}  out-grad rel "
                  f"{rel_l2(got[k], to_ndhwc(kept[k].grad)):.2e}  shape {tuple(kept[k].shape)}")
    Pn = dict(net.named_parameters())
    for k in sorted(P):
        if P[k].requires_grad and P[k].grad is not None:
            print(f"  grad {k:34s} {rel_l2(Pn[k].grad, P[k].grad):.2e}")


main()
