"""R(2+1)D-18 video backbone (``torchvision.models.video.r2plus1d_18``, used by the reference's
``VidResNet``, src/models/frame_transformer.py:64-74) on HIP kernels.  SURVEY section 8 row a11.

torchvision is not installed in the build image and its pretrained weights need the network, so
this module restates the public architecture (same module tree => same state-dict keys:
``stem.{0,1,3,4}``, ``layer{1..4}.{i}.conv{1,2}.0.{0,1,3}``, ``...conv{1,2}.1``,
``...downsample.{0,1}``, ``fc``); parity against torchvision itself is UNPINNED (no golden vectors
can be produced here) and is checked against a torch-CPU conv3d restatement only.

Layout: activations are NDHWC matrices [(n t h w), C].  The factorised convolutions map onto the
2-D path: (1,k,k) spatial = a 2-D conv over the N*T frames; (3,1,1) temporal = a (3,1) conv over
the [T, H*W] view of each clip; the strided 1x1x1 downsample = two strided row gathers + a GEMM.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as F
from .. import ops


class Conv2Plus1D(nn.Sequential):
    def __init__(self, in_planes, out_planes, midplanes, stride=1, padding=1):
        super().__init__(
            nn.Conv3d(in_planes, midplanes, kernel_size=(1, 3, 3), stride=(1, stride, stride),
                      padding=(0, padding, padding), bias=False),
            nn.BatchNorm3d(midplanes),
            nn.ReLU(inplace=True),
            nn.Conv3d(midplanes, out_planes, kernel_size=(3, 1, 1), stride=(stride, 1, 1),
                      padding=(padding, 0, 0), bias=False),
        )

    @staticmethod
    def get_downsample_stride(stride):
        return stride, stride, stride


CPAD = 64     # mid-plane counts 45 / 230 / 460 / 921 are zero-padded to multiples of 64 (MFMA / LDS-DMA tile widths)


def _mid_cpad(conv, frames, H, W, dtype):
    """Channel padding of a spatial convolution's mid planes.  Multiples of 32 (288, 576, 1152) are whole k-tiles per filter
    tap for every consumer as they are; layer 1's 144 stay 144 where the streamed-weight halo kernels take the layer and its
    data gradient (ops.conv3x3_stream: the temporal convolution behind reads 144-channel pixels through the implicit
    kernels' per-lane taps) -- 192 would be a third more bytes and MFMA work on every mid-plane map; the rest is padded to
    the 64-wide tiles."""
    c = conv.out_channels
    if dtype not in (torch.bfloat16, torch.float16):
        return CPAD
    if c % 32 == 0:
        return 32
    if (c == 144 and conv.in_channels == 64 and tuple(conv.kernel_size[1:]) == (3, 3) and tuple(conv.stride[1:]) == (1, 1)
            and tuple(conv.padding[1:]) == (1, 1) and ops.conv3x3_stream_geometry(frames, H, W, 144, 64, dtype)
            and ops.conv3x3_stream_geometry(frames, H, W, 64, 144, dtype)):
        return 16
    return CPAD


def _spatial(fm, conv, bn, relu, dtype, fork=None, defer=False):
    """(1,k,k) conv + BN(+ReLU) on an NDHWC matrix: 2-D conv over N*T frames.  fork="alias": the map has a second consumer
    (the block's shortcut); the layer hands it out as a second result so that the shortcut's gradient joins this layer's
    data gradient inside the kernel that writes it (F._ConvBnAct) -> (fm, alias).  defer: the BatchNorm (+ ReLU) is NOT
    applied here -- the map returned is the convolution's output z and the third result the affine the temporal half behind
    applies inside its window kernels (F.WINDOW_VIRTUAL_BN) -> (fm, alias or None, affine)."""
    y, N, T, H, W = fm
    k, s, p = conv.kernel_size[1:], conv.stride[1:], conv.padding[1:]
    Ho, Wo = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
    cpad = _mid_cpad(conv, N * T, H, W, dtype)
    if defer:
        out, affine = F.conv_bn_act_raw(y, conv.weight, bn, (N * T, y.shape[1], H, W, False), k, s, p, relu=relu, dtype=dtype,
                                        cpad=cpad, fork=fork, defer_apply=True)
        if fork is not None:
            return (out[0], N, T, Ho, Wo), out[1], affine
        return (out, N, T, Ho, Wo), None, affine
    out = F.conv_bn_act_raw(y, conv.weight, bn, (N * T, y.shape[1], H, W, False), k, s, p, relu=relu, dtype=dtype,
                            cpad=cpad, fork=fork)
    if fork is not None:
        return (out[0], N, T, Ho, Wo), out[1]
    return (out, N, T, Ho, Wo)


def _temporal(fm, conv, bn, relu, dtype, residual=None, in_affine=None):
    """(3,1,1) conv + BN(+res)(+ReLU): a (kt,1) conv over the [T, H*W] view of each clip.  in_affine: fm holds the spatial
    half's convolution output and this is the BatchNorm (+ ReLU) between the halves (see _spatial(defer=True))."""
    y, N, T, H, W = fm
    kt, st, pt = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    out = F.conv_bn_act_raw(y, conv.weight, bn, (N, y.shape[1], T, H * W, False), (kt, 1), (st, 1), (pt, 0),
                            relu=relu, residual=residual, dtype=dtype, cpad=CPAD, in_affine=in_affine)
    To = (T + 2 * pt - kt) // st + 1
    return (out, N, To, H, W)


def _virtual_bn_pair(pair, N, T, H, W, dtype) -> bool:
    """Can the BatchNorm + ReLU between the two halves of this Conv2Plus1D stay virtual?  Needs the window kernels of the
    temporal half (144 mid planes -> 64, (3, 1, 1) / 1 / pad 1, a segment length that fits) and un-padded mid planes."""
    sp, tm = pair[0], pair[3]
    if not (F.WINDOW_VIRTUAL_BN and F.WINDOW_FWD and F.HALO_CONV and dtype in (torch.bfloat16, torch.float16)):
        return False
    if not (sp.out_channels == 144 and tm.in_channels == 144 and tm.out_channels == 64 and tuple(tm.kernel_size) == (3, 1, 1)
            and tuple(tm.stride) == (1, 1, 1) and tuple(tm.padding) == (1, 0, 0) and tuple(sp.stride) == (1, 1, 1)):
        return False
    if _mid_cpad(sp, N * T, H, W, dtype) != 16:           # (144 stays 144 only on the streamed-weight path)
        return False
    return ops.conv3x1_window_geometry(N, T, H * W, 144, 64, dtype)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, conv_builder, stride=1, downsample=None):
        midplanes = (inplanes * planes * 3 * 3 * 3) // (inplanes * 3 * 3 + 3 * planes)
        super().__init__()
        self.conv1 = nn.Sequential(conv_builder(inplanes, planes, midplanes, stride), nn.BatchNorm3d(planes),
                                   nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(conv_builder(planes, planes, midplanes), nn.BatchNorm3d(planes))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_ndhwc(self, fm, dtype):
        y, N, T, H, W = fm
        c1, c2 = self.conv1[0], self.conv2[0]
        if _virtual_bn_pair(c1, N, T, H, W, dtype):
            # the spatial half leaves its output z; the BatchNorm + ReLU between the halves runs inside the temporal half's kernels
            out, y, aff = _spatial(fm, c1[0], c1[1], True, dtype, fork="alias", defer=True)
            out = _temporal(out, c1[3], self.conv1[1], True, dtype, in_affine=aff)
        else:
            out, y = _spatial(fm, c1[0], c1[1], True, dtype, fork="alias")      # y: the block input again, for the shortcut
            out = _temporal(out, c1[3], self.conv1[1], True, dtype)
        residual = y
        if self.downsample is not None:
            ds, dbn = self.downsample[0], self.downsample[1]
            st = ds.stride
            C = ds.in_channels
            r = F.subsample_nhwc(y, N * T, C, H, W, (st[1], st[2]))                       # spatial stride
            Hs, Ws = (H - 1) // st[1] + 1, (W - 1) // st[2] + 1
            r = F.subsample_nhwc(r, N, C, T, Hs * Ws, (st[0], 1))                        # temporal stride
            Ts = (T - 1) // st[0] + 1
            residual = F.conv_bn_act_raw(r, ds.weight, dbn, (N * Ts, C, Hs, Ws, False), 1, 1, 0, relu=False, dtype=dtype)
        _, N2, T2, H2, W2 = out
        if _virtual_bn_pair(c2, N2, T2, H2, W2, dtype):
            out, _, aff = _spatial(out, c2[0], c2[1], True, dtype, defer=True)
            return _temporal(out, c2[3], self.conv2[1], True, dtype, residual=residual, in_affine=aff)
        out = _spatial(out, c2[0], c2[1], True, dtype)
        return _temporal(out, c2[3], self.conv2[1], True, dtype, residual=residual)     # out += residual; relu


class R2Plus1dStem(nn.Sequential):
    def __init__(self):
        super().__init__(
            nn.Conv3d(3, 45, kernel_size=(1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3), bias=False),
            nn.BatchNorm3d(45), nn.ReLU(inplace=True),
            nn.Conv3d(45, 64, kernel_size=(3, 1, 1), stride=(1, 1, 1), padding=(1, 0, 0), bias=False),
            nn.BatchNorm3d(64), nn.ReLU(inplace=True))


class VideoResNet(nn.Module):
    def __init__(self, layers=(2, 2, 2, 2), num_classes=400, *, compute_dtype=torch.bfloat16):
        super().__init__()
        self.inplanes = 64
        self.stem = R2Plus1dStem()
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Linear(512, num_classes)
        self.compute_dtype = compute_dtype
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            ds = Conv2Plus1D.get_downsample_stride(stride)
            downsample = nn.Sequential(nn.Conv3d(self.inplanes, planes, kernel_size=1, stride=ds, bias=False),
                                       nn.BatchNorm3d(planes))
        layers = [BasicBlock(self.inplanes, planes, Conv2Plus1D, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes, Conv2Plus1D))
        return nn.Sequential(*layers)

    def features(self, x):
        """x [N, 3, T, H, W] -> pooled [N, 512]."""
        if x.dim() != 5 or x.shape[1] != 3:
            raise ValueError("VideoResNet expects clips [N, 3, T, H, W]")
        dt = self.compute_dtype
        N, _, T, H, W = x.shape
        frames = x.permute(0, 2, 1, 3, 4)                  # [N, T, 3, H, W]: per-frame NCHW
        if not frames.is_contiguous():
            frames = frames.contiguous()
        s0, b0, s3, b3 = self.stem[0], self.stem[1], self.stem[3], self.stem[4]
        k, s, p = s0.kernel_size[1:], s0.stride[1:], s0.padding[1:]
        # input_grad_clips: indices of the clips whose pixel gradient is consumed (None = all): the caller's hint that
        # only the learnable CLS clip of each sample needs d(loss)/d(pixels)
        hint = getattr(self, "input_grad_clips", None)
        dx_frames = None if hint is None else [(int(c) * T, T) for c in hint]
        y = F.conv_bn_act_raw(frames.view(N * T, 3, H, W), s0.weight, b0, (N * T, 3, H, W, True), k, s, p, relu=True, dtype=dt,
                              cpad=CPAD, dx_frames=dx_frames)
        H1, W1 = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
        fm = _temporal((y, N, T, H1, W1), s3, b3, True, dt)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                fm = blk.forward_ndhwc(fm, dt)
        y, N, T, H, W = fm
        return F.mean_rows(y.view(N, T * H * W, y.shape[1]))          # AdaptiveAvgPool3d(1)

    def forward(self, x):
        feats = self.features(x)
        fc = self.fc[0] if isinstance(self.fc, nn.Sequential) else self.fc
        return F.linear(feats, fc.weight, fc.bias)


def r2plus1d_18(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained=True downloads Kinetics weights (torchvision); there is no network here -- "
                           "load a state_dict explicitly")
    return VideoResNet((2, 2, 2, 2), **kwargs)
