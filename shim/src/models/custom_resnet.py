"""Drop-in for the reference's ``src/models/custom_resnet.py:19-207``: the same names, served by the MI355X build."""
from dvt_amd.models.custom_resnet import conv3x3, BasicBlock, Bottleneck, ResNet, resnet18, resnet34, resnet50, resnet101, resnet152  # noqa: F401

__all__ = ['conv3x3', 'BasicBlock', 'Bottleneck', 'ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152']
