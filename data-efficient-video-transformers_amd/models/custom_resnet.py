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


def _cba(fm: FMap, conv, bn, relu, residual=None, dtype=torch.bfloat16) -> FMap:
    x, N, H, W = fm
    y = F.conv_bn_act(x, conv, bn, (N, conv.in_channels, H, W, False), relu=relu, residual=residual, dtype=dtype)
    Ho, Wo = _out_hw(conv, H, W)
    return (y, N, Ho, Wo)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward_nhwc(self, fm: FMap, dtype) -> FMap:
        residual = fm[0]
        out = _cba(fm, self.conv1, self.bn1, True, dtype=dtype)
        if self.downsample is not None:
            residual = _cba(fm, self.downsample[0], self.downsample[1], False, dtype=dtype)[0]
        return _cba(out, self.conv2, self.bn2, True, residual=residual, dtype=dtype)   # out += residual; relu


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_nhwc(self, fm: FMap, dtype) -> FMap:
        residual = fm[0]
        out = _cba(fm, self.conv1, self.bn1, True, dtype=dtype)
        out = _cba(out, self.conv2, self.bn2, True, dtype=dtype)
        if self.downsample is not None:
            residual = _cba(fm, self.downsample[0], self.downsample[1], False, dtype=dtype)[0]
        return _cba(out, self.conv3, self.bn3, True, residual=residual, dtype=dtype)


class ResNet(nn.Module):

    def __init__(self, block, layers, num_classes=1000, *, compute_dtype=torch.bfloat16):
        self.inplanes = 64
        super(ResNet, self).__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        self.compute_dtype = compute_dtype

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward_nhwc(self, x) -> List[FMap]:
        """x [N, 3, H, W] (NCHW frames) -> [(x2), (x3), (x4)] as NHWC matrices."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("ResNet expects frames [N, 3, H, W]")
        dt = self.compute_dtype
        N, _, H, W = x.shape
        y = F.conv_bn_act(x, self.conv1, self.bn1, (N, 3, H, W, True), relu=True, dtype=dt)   # stem reads NCHW
        H1, W1 = _out_hw(self.conv1, H, W)
        mp = self.maxpool
        y = F.maxpool_nhwc(y, N, 64, H1, W1, mp.kernel_size, mp.stride, mp.padding)
        H2 = (H1 + 2 * mp.padding - mp.kernel_size) // mp.stride + 1
        W2 = (W1 + 2 * mp.padding - mp.kernel_size) // mp.stride + 1
        fm: FMap = (y, N, H2, W2)
        outs = []
        for li, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            for blk in layer:
                fm = blk.forward_nhwc(fm, dt)
            if li >= 1:
                outs.append(fm)
        return outs

    def forward(self, x):
        outs = []
        for (y, N, H, W) in self.forward_nhwc(x):
            C = y.shape[1]
            outs.append(F.transpose_last2(y.view(N, H * W, C)).view(N, C, H, W))     # NHWC -> NCHW
        return tuple(outs)                                                           # (x2, x3, x4)


def _no_pretrained(flag):
    if flag:
        raise RuntimeError("pretrained=True downloads ImageNet weights (model_zoo, custom_resnet.py:162-163); "
                           "there is no network here -- load a state_dict explicitly")


def resnet18(pretrained=False, **kwargs):
    _no_pretrained(pretrained)
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34(pretrained=False, **kwargs):
    _no_pretrained(pretrained)
    return ResNet(BasicBlock, [3, 4, 6, 3], **kwargs)


def resnet50(pretrained=False, **kwargs):
    _no_pretrained(pretrained)
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101(pretrained=False, **kwargs):
    _no_pretrained(pretrained)
    return ResNet(Bottleneck, [3, 4, 23, 3], **kwargs)


def resnet152(pretrained=False, **kwargs):
    _no_pretrained(pretrained)
    return ResNet(Bottleneck, [3, 8, 36, 3], **kwargs)
