"""Mirror of the reference's ``src/models/custom_resnet.py`` (torchvision-style ResNet whose
``forward`` returns the 3-scale pyramid ``(x2, x3, x4)`` = 128x28^2, 256x14^2, 512x7^2 at 224^2
input) on HIP kernels.  SURVEY section 8 row a12.

Kept: ``conv3x3``, ``BasicBlock``, ``Bottleneck``, ``ResNet(block, layers, num_classes)``,
``resnet18/34/50/101/152(pretrained=False)``, attribute names and therefore the torchvision
state-dict keys (``conv1.weight``, ``bn1.*``, ``layer{1..4}.{i}.conv{1,2}.weight``,
``...downsample.{0,1}.*``, ``fc.*``), He-normal conv init / BN (1, 0) init
(custom_resnet.py:113-119).  ``nn.Conv2d`` / ``nn.BatchNorm2d`` are parameter containers; the
arithmetic runs as im2col -> MFMA GEMM -> BatchNorm(+residual)+ReLU on NHWC feature maps.

Deviations: the reference also computes ``avgpool(7) -> fc`` and throws the result away
(custom_resnet.py:149-153); that dead computation (and the fixed 7x7 pool that makes the
reference fail for inputs other than ~224^2) is not executed, ``fc`` stays in the state dict.
``pretrained=True`` needs the network (model_zoo) and raises.  ``forward`` returns NCHW tensors
like the reference; ``forward_nhwc`` returns the internal NHWC matrices (used by TPN).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn as nn

from .. import functional as F

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152']

FMap = Tuple[torch.Tensor, int, int, int]      # (NHWC matrix [N*H*W, C], N, H, W)


def conv3x3(in_planes, out_planes, stride=1):
    """3x3 convolution with padding (custom_resnet.py:19-22)."""
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _out_hw(conv: nn.Conv2d, H: int, W: int):
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    return (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1


def _cba(fm: FMap, conv, bn, relu, residual=None, dtype=torch.bfloat16, fork=None, stride=None):
    """conv -> bn (-> + residual) (-> relu) on an NHWC map.  fork: the map has a second consumer (the block's shortcut) and the
    layer hands it out as a second result (see F._ConvBnAct): -> (FMap, second).  stride: overrides the module's (a strided
    1x1 downsample convolution applied to an already subsampled map runs with stride 1)."""
    x, N, H, W = fm
    y = F.conv_bn_act(x, conv, bn, (N, conv.in_channels, H, W, False), relu=relu, residual=residual, dtype=dtype, fork=fork,
                      stride=stride)
    Ho, Wo = _out_hw(conv, H, W) if stride is None else (H, W)
    if fork is not None:
        return (y[0], N, Ho, Wo), y[1]
    return (y, N, Ho, Wo)


class _ResidualBlock(nn.Module):
    """conv{i} / bn{i} pairs (i = 1..len(spec)) + optional ``downsample``; the last pair takes the shortcut.
    ``spec``: per convolution (kernel, out_planes, stride).  Attribute names are the torchvision ones, so the
    state-dict keys match the reference's (custom_resnet.py:25-93)."""
    expansion = 1

    def _build(self, inplanes, spec, downsample, stride):
        cin = inplanes
        for i, (k, cout, s) in enumerate(spec, start=1):
            conv = conv3x3(cin, cout, s) if k == 3 else nn.Conv2d(cin, cout, kernel_size=1, stride=s, bias=False)
            setattr(self, f"conv{i}", conv)
            setattr(self, f"bn{i}", nn.BatchNorm2d(cout))
            cin = cout
        self.depth = len(spec)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_nhwc(self, fm: FMap, dtype) -> FMap:
        # The block input has two consumers, conv1 and the shortcut.  conv1's Function hands the shortcut its input as a second
        # result, so that the shortcut's gradient comes back to it and joins conv1's data gradient inside the kernel that writes
        # it (no add kernel).  A strided 1x1 downsample convolution gets the SUBSAMPLED map (its own gather) and runs with
        # stride 1 on it: its input gradient then returns compact instead of as a zero-filled full-size map.
        ds = self.downsample
        fork = "alias"
        if ds is not None:
            dconv = ds[0]
            s_ = dconv.stride[0]
            if dconv.kernel_size == (1, 1) and dconv.padding == (0, 0) and dconv.stride == (s_, s_) and s_ > 1:
                fork = s_
        out, second = _cba(fm, self.conv1, self.bn1, True, dtype=dtype, fork=fork)
        _, N, H, W = fm
        if ds is None:
            shortcut = second
        elif fork == "alias":
            shortcut = _cba((second, N, H, W), ds[0], ds[1], False, dtype=dtype)[0]
        else:
            Hs, Ws = (H + fork - 1) // fork, (W + fork - 1) // fork
            shortcut = _cba((second, N, Hs, Ws), ds[0], ds[1], False, dtype=dtype, stride=(1, 1))[0]
        for i in range(2, self.depth):
            out = _cba(out, getattr(self, f"conv{i}"), getattr(self, f"bn{i}"), True, dtype=dtype)
        last = self.depth                                     # out += residual; relu  (fused into the BatchNorm pass)
        return _cba(out, getattr(self, f"conv{last}"), getattr(self, f"bn{last}"), True, residual=shortcut, dtype=dtype)


class BasicBlock(_ResidualBlock):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self._build(inplanes, [(3, planes, stride), (3, planes, 1)], downsample, stride)


class Bottleneck(_ResidualBlock):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self._build(inplanes, [(1, planes, 1), (3, planes, stride), (1, planes * 4, 1)], downsample, stride)


class ResNet(nn.Module):
    _STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))          # (planes, stride of the first block) of layer1..4

    def __init__(self, block, layers, num_classes=1000, *, compute_dtype=torch.bfloat16):
        super().__init__()
        self.inplanes = 64
        self.compute_dtype = compute_dtype
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, ((planes, stride), blocks) in enumerate(zip(self._STAGES, layers), start=1):
            setattr(self, f"layer{i}", self._make_layer(block, planes, blocks, stride))
        self.avgpool = nn.AvgPool2d(7, stride=1)                # kept for the state dict / attribute surface only
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        self._reset_parameters()

    def _reset_parameters(self):
        """He-normal convolutions (fan-out), unit BatchNorm (custom_resnet.py:113-119)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _make_layer(self, block, planes, blocks, stride=1):
        width = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != width:
            shortcut = nn.Sequential(nn.Conv2d(self.inplanes, width, kernel_size=1, stride=stride, bias=False),
                                     nn.BatchNorm2d(width))
        stage = [block(self.inplanes, planes, stride, shortcut)]
        self.inplanes = width
        stage += [block(width, planes) for _ in range(blocks - 1)]
        return nn.Sequential(*stage)

    def forward_nhwc(self, x) -> List[FMap]:
        """x [N, 3, H, W] (NCHW frames) -> [(x2), (x3), (x4)] as NHWC matrices."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("ResNet expects frames [N, 3, H, W]")
        dt = self.compute_dtype
        N, _, H, W = x.shape
        mp = self.maxpool
        fused_pool = (mp.kernel_size, mp.stride, mp.padding) == (3, 2, 1)     # bn1 -> relu -> maxpool in one pass
        y = F.conv_bn_act(x, self.conv1, self.bn1, (N, 3, H, W, True), relu=True, dtype=dt, pool=fused_pool)   # stem reads NCHW
        H1, W1 = _out_hw(self.conv1, H, W)
        if not fused_pool:
            y = F.maxpool_nhwc(y, N, 64, H1, W1, mp.kernel_size, mp.stride, mp.padding)
        H2 = (H1 + 2 * mp.padding - mp.kernel_size) // mp.stride + 1
        W2 = (W1 + 2 * mp.padding - mp.kernel_size) // mp.stride + 1
        fm: FMap = (y, N, H2, W2)
        pyramid = []
        for i in range(1, 5):
            for blk in getattr(self, f"layer{i}"):
                fm = blk.forward_nhwc(fm, dt)
            if i >= 2:
                if i < 4:                                       # x2 / x3 also feed the next stage
                    keep, cont = F.fork(fm[0])
                    pyramid.append((keep,) + tuple(fm[1:]))
                    fm = (cont,) + tuple(fm[1:])
                else:
                    pyramid.append(fm)
        return pyramid

    def forward(self, x):
        maps = []
        for (y, N, H, W) in self.forward_nhwc(x):
            C = y.shape[1]
            maps.append(F.transpose_last2(y.view(N, H * W, C)).view(N, C, H, W))     # NHWC -> NCHW
        return tuple(maps)                                                           # (x2, x3, x4)


_DEPTHS = {"resnet18": (BasicBlock, (2, 2, 2, 2)), "resnet34": (BasicBlock, (3, 4, 6, 3)),
           "resnet50": (Bottleneck, (3, 4, 6, 3)), "resnet101": (Bottleneck, (3, 4, 23, 3)),
           "resnet152": (Bottleneck, (3, 8, 36, 3))}


def _factory(name):
    block, layers = _DEPTHS[name]

    def make(pretrained=False, **kwargs):
        if pretrained:
            raise RuntimeError("pretrained=True downloads ImageNet weights (model_zoo, custom_resnet.py:162-163); "
                               "there is no network here -- load a state_dict explicitly")
        return ResNet(block, list(layers), **kwargs)
    make.__name__ = name
    make.__doc__ = f"{name} returning the (x2, x3, x4) pyramid (custom_resnet.py:156-211)."
    return make


resnet18, resnet34, resnet50, resnet101, resnet152 = (_factory(n) for n in _DEPTHS)
