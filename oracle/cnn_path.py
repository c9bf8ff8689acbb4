"""CPU oracle (TEST INFRASTRUCTURE) for the per-frame CNN encoder and the pyramid
(src/models/custom_resnet.py, src/models/TPN.py).  Functional restatement over a reference
state dict; the convolution / batch-norm / max-pool arithmetic is torch's own CPU
implementation, which is what the reference delegates to.  Pinned by
tests/test_oracle_golden.py against tests/golden/resnet18_pyramid.npz (generated from the
imported reference)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as TF

Tensor = torch.Tensor


def _bn(x: Tensor, P: Dict[str, Tensor], prefix: str, training: bool, stats: dict) -> Tensor:
    """nn.BatchNorm2d (eps 1e-5, momentum 0.1).  In training mode the updated running statistics are
    returned through ``stats`` (functional, the inputs are not modified)."""
    rm, rv = P[prefix + "running_mean"].clone(), P[prefix + "running_var"].clone()
    y = TF.batch_norm(x, rm, rv, P[prefix + "weight"], P[prefix + "bias"], training, 0.1, 1e-5)
    stats[prefix] = (rm, rv)
    return y


def basic_block(x: Tensor, P, prefix: str, stride: int, training: bool, stats: dict) -> Tensor:
    """BasicBlock.forward, custom_resnet.py:38-54."""
    out = TF.conv2d(x, P[prefix + "conv1.weight"], None, stride, 1)
    out = torch.relu(_bn(out, P, prefix + "bn1.", training, stats))
    out = TF.conv2d(out, P[prefix + "conv2.weight"], None, 1, 1)
    out = _bn(out, P, prefix + "bn2.", training, stats)
    residual = x
    if prefix + "downsample.0.weight" in P:
        residual = TF.conv2d(x, P[prefix + "downsample.0.weight"], None, stride, 0)
        residual = _bn(residual, P, prefix + "downsample.1.", training, stats)
    return torch.relu(out + residual)


def resnet_pyramid(x: Tensor, P: Dict[str, Tensor], layers: List[int], training: bool = True
                   ) -> Tuple[Tensor, Tensor, Tensor, dict]:
    """ResNet.forward with BasicBlocks, custom_resnet.py:138-153: returns (x2, x3, x4, running stats);
    the discarded avgpool+fc tail is omitted."""
    stats = {}
    x = TF.conv2d(x, P["conv1.weight"], None, 2, 3)
    x = torch.relu(_bn(x, P, "bn1.", training, stats))
    x = TF.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, nb in enumerate(layers):
        for b in range(nb):
            stride = 2 if (li > 0 and b == 0) else 1
            x = basic_block(x, P, f"layer{li + 1}.{b}.", stride, training, stats)
        feats.append(x)
    return feats[1], feats[2], feats[3], stats


# --------------------------------------------------------------------------
# Pyramid head and temporal grouping (src/models/TPN.py)
# --------------------------------------------------------------------------
def pyramid_vector(feat: Tensor, w=None, b=None) -> Tensor:
    """Feature_Pyramid_*: global AvgPool2d over the whole map (k = 28 / 14 / 7, TPN.py:6,20,33) then an
    optional 1x1 convolution on the pooled vector (TPN.py:8,35); ``high`` has none (TPN.py:24-26)."""
    v = feat.mean(dim=(2, 3))
    if w is not None:
        v = v @ w.reshape(w.shape[0], -1).t() + b
    return v


def reasoning(x: Tensor, P: Dict[str, Tensor], start: int = 2, max_group: int = 4) -> Tensor:
    """Reasoning.forward (TPN.py:106-112), eval mode: for g in 2..4: sum_group -> ReLU -> Linear ->
    ReLU -> Linear -> ReLU -> Linear -> Sigmoid; mean of the three predictions."""
    from .clip_path import sum_group, linear
    pred = 0
    for g in range(start, max_group + 1):
        p = f"relation.{g - start}."
        s = sum_group(x, g)
        s = linear(torch.relu(s), P[p + "1.weight"], P[p + "1.bias"])
        s = linear(torch.relu(s), P[p + "4.weight"], P[p + "4.bias"])
        s = torch.sigmoid(linear(torch.relu(s), P[p + "7.weight"], P[p + "7.bias"]))
        pred = pred + s
    return pred / (max_group - start + 1)


# --------------------------------------------------------------------------
# R(2+1)D-18 (torchvision.models.video.r2plus1d_18, used at frame_transformer.py:67).
# Restated from the public architecture; NOT pinned against torchvision (not installed here).
# --------------------------------------------------------------------------
def _bn3(x, P, prefix, training):
    return TF.batch_norm(x, P[prefix + "running_mean"].clone(), P[prefix + "running_var"].clone(),
                         P[prefix + "weight"], P[prefix + "bias"], training, 0.1, 1e-5)


def _conv2plus1d(x, P, prefix, stride, training):
    x = TF.conv3d(x, P[prefix + "0.weight"], None, (1, stride, stride), (0, 1, 1))
    x = torch.relu(_bn3(x, P, prefix + "1.", training))
    return TF.conv3d(x, P[prefix + "3.weight"], None, (stride, 1, 1), (1, 0, 0))


def r2plus1d_features(x: Tensor, P: Dict[str, Tensor], layers=(2, 2, 2, 2), training: bool = True) -> Tensor:
    """x [N, 3, T, H, W] -> pooled [N, 512]."""
    x = TF.conv3d(x, P["stem.0.weight"], None, (1, 2, 2), (0, 3, 3))
    x = torch.relu(_bn3(x, P, "stem.1.", training))
    x = TF.conv3d(x, P["stem.3.weight"], None, 1, (1, 0, 0))
    x = torch.relu(_bn3(x, P, "stem.4.", training))
    for li, nb in enumerate(layers):
        for b in range(nb):
            pre = f"layer{li + 1}.{b}."
            stride = 2 if (li > 0 and b == 0) else 1
            out = _conv2plus1d(x, P, pre + "conv1.0.", stride, training)
            out = torch.relu(_bn3(out, P, pre + "conv1.1.", training))
            out = _conv2plus1d(out, P, pre + "conv2.0.", 1, training)
            out = _bn3(out, P, pre + "conv2.1.", training)
            res = x
            if pre + "downsample.0.weight" in P:
                res = TF.conv3d(x, P[pre + "downsample.0.weight"], None, stride)
                res = _bn3(res, P, pre + "downsample.1.", training)
            x = torch.relu(out + res)
    return x.mean(dim=(2, 3, 4))
