"""Mirror of the reference's ``src/models/TPN.py`` ("spatial-temporal pyramid"): the three CNN
scales are globally average-pooled, the low / mid ones pass a 1x1 convolution, the three vectors
are concatenated to one 896-d token per frame, and ``Reasoning`` applies multi-scale temporal
grouping (sums of 2 / 3 / 4 consecutive frames) + MLPs + sigmoid, averaged.  SURVEY section 8
rows a13, a14.

Kept: class names, attribute names (state-dict keys ``net.*``, ``pyramid_{low,mid,high}.channels_reduce.*``,
``reason.relation.{0,1,2}.{1,4,7}.*``), constructor defaults.  The reference file has no import
statements (NameError on import) and builds ``custom_resnet.resnet34(True)`` (pretrained download);
here the backbone is ``resnet34(False)`` unless a state dict is loaded.  ``Feature_Pyramid_High`` keeps
its unused 1x1 convolution (TPN.py:22) in the state dict.  Dropout (0.6 / 0.5, TPN.py:91,94) draws its masks from
the Philox dropout kernel in training mode (torch's generator stream cannot be reproduced); parity is defined in eval mode.

On NHWC matrices a global AvgPool2d(k) over a k x k map is a mean over rows and a 1x1 convolution on
the pooled vector is a Linear, so the pyramid head is two tiny GEMMs.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as F
from ..lightning_compat import LightningModule
from . import custom_resnet


def _pool(fm):
    """Global average over the spatial positions of an NHWC feature matrix -> [N, C]."""
    y, N, H, W = fm
    return F.mean_rows(y.view(N, H * W, y.shape[1]))


def _conv1x1(vec, conv: nn.Conv2d):
    return F.linear(vec, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)


class Feature_Pyramid_Mid(nn.Module):
    def __init__(self):
        super(Feature_Pyramid_Mid, self).__init__()
        self.pool_branch = nn.Sequential(nn.AvgPool2d(kernel_size=14))
        self.channels_reduce = nn.Conv2d(256, 256, kernel_size=1)

    def forward(self, mid):
        return _conv1x1(_pool(mid), self.channels_reduce)


class Feature_Pyramid_High(nn.Module):
    def __init__(self):
        super(Feature_Pyramid_High, self).__init__()
        self.pool_branch = nn.Sequential(nn.AvgPool2d(kernel_size=7))
        self.channels_reduce = nn.Conv2d(512, 512, kernel_size=1)      # unused in forward (TPN.py:24-26)

    def forward(self, high):
        return _pool(high)


class Feature_Pyramid_low(nn.Module):
    def __init__(self):
        super(Feature_Pyramid_low, self).__init__()
        self.pool_branch = nn.Sequential(nn.AvgPool2d(kernel_size=28))
        self.channels_reduce = nn.Conv2d(128, 128, kernel_size=1)

    def forward(self, low):
        return _conv1x1(_pool(low), self.channels_reduce)


def sum_group(x, groups=2):
    """TPN.py:64-72: x [batch, pics, vector] -> sums of `groups` consecutive frames, concatenated."""
    batch, pics, vector = x.shape
    g = pics // groups
    xs = x[:, : g * groups].contiguous().view(batch * g, groups, vector)
    return F.sum_rows(xs).view(batch, g * vector)


class Reasoning(nn.Module):
    def __init__(self, num_segments=4, num_frames=5, num_class=15, img_dim=896, max_group=4, start=2):
        super(Reasoning, self).__init__()
        self.num_segments = num_segments
        self.num_frames = num_frames
        self.num_class = num_class
        self.img_feature_dim = img_dim
        self.num_groups = max_group
        self.start = start
        self.relation = nn.ModuleList()
        self.classifier_scales = nn.ModuleList()
        num_bottleneck = 512
        for scales in range(self.start, self.num_groups + 1):
            fc_fusion = nn.Sequential(
                nn.ReLU(),
                nn.Linear(self.img_feature_dim * int(self.num_segments * self.num_frames / scales), num_bottleneck),
                nn.ReLU(),
                nn.Dropout(p=0.6),
                nn.Linear(num_bottleneck, num_bottleneck),
                nn.ReLU(),
                nn.Dropout(p=0.5),
                nn.Linear(num_bottleneck, self.num_class),
                nn.Sigmoid(),
            )
            self.relation += [fc_fusion]

    def forward(self, x):
        prediction = None
        for segment_group in range(self.start, self.num_groups + 1):
            net = self.relation[segment_group - self.start]
            s = sum_group(x, groups=segment_group)
            s = F.linear(F.relu(s), net[1].weight, net[1].bias)
            s = F.linear(F.dropout(F.relu(s), net[3].p, self.training), net[4].weight, net[4].bias)       # TPN.py:91-93
            s = F.sigmoid(F.linear(F.dropout(F.relu(s), net[6].p, self.training), net[7].weight, net[7].bias))  # :94-97
            prediction = s if prediction is None else F.add(prediction, s)
        return F.scale_f32(prediction, 1.0 / (self.num_groups - self.start + 1))


class TPN(LightningModule):
    def __init__(self, compute_dtype=torch.bfloat16):
        super(TPN, self).__init__()
        self.net = custom_resnet.resnet34(False, compute_dtype=compute_dtype)      # reference: resnet34(True)
        self.pyramid_low = Feature_Pyramid_low()
        self.pyramid_mid = Feature_Pyramid_Mid()
        self.pyramid_high = Feature_Pyramid_High()
        self.reason = Reasoning()

    def frame_tokens(self, x):
        """x [frames, 3, H, W] -> one 896-d token per frame: cat(high, mid, low) (TPN.py:54-58)."""
        low, mid, high = self.net.forward_nhwc(x)
        return F.concat_cols(self.pyramid_high(high), self.pyramid_mid(mid), self.pyramid_low(low))

    def forward(self, x):
        cnn_out = self.frame_tokens(x)                     # [frames, 896]
        return self.reason(cnn_out.unsqueeze(0))           # the reference treats the frames as one video
