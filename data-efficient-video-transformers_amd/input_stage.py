"""On-device input stage: decoded uint8 frames -> normalised clip tensors.

Mirrors the four ``transforms.Compose`` pipelines of the reference loader
(src/dataloaders/mmx/MMX_Light_dl.py:184-217) for inputs that are already decoded RGB arrays in HBM; the
random training augmentations of the *image* branch (RandomResizedCrop / flips / AutoAugment, :185-188) are
host-side policy and stay outside (the image branch is disabled in the reference's ``__getitem__`` anyway, :276).
"""
from __future__ import annotations

import torch

from . import ops

KINETICS_MEAN, KINETICS_STD = (0.43216, 0.394666, 0.37645), (0.22803, 0.22145, 0.216989)     # :206-207
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)                    # :191-192


class ClipPreprocessor:
    """``Resize(resize) -> CenterCrop(crop) -> ToTensor -> Normalize(mean, std)`` for whole batches of frames.

    ``__call__(frames)``: uint8 ``[..., H0, W0, 3]`` (any leading dims, e.g. ``[B, 13, 12, H0, W0, 3]``) ->
    ``[..., 3, crop, crop]`` in ``dtype`` -- the layout ``FrameTransformer.vid_step`` / ``ViViT.forward`` consume.
    """

    def __init__(self, resize: int, crop: int, mean=KINETICS_MEAN, std=KINETICS_STD, dtype: torch.dtype = torch.bfloat16):
        self.resize, self.crop, self.mean, self.std, self.dtype = resize, crop, tuple(mean), tuple(std), dtype

    def __call__(self, frames: torch.Tensor) -> torch.Tensor:
        lead = frames.shape[:-3]
        flat = frames.reshape(-1, *frames.shape[-3:])
        out = ops.frames_preprocess(flat, self.resize, self.crop, self.mean, self.std, self.dtype)
        return out.view(*lead, 3, self.crop, self.crop)


def train_vid(dtype=torch.bfloat16):      # MMX_Light_dl.py:203-208
    return ClipPreprocessor(120, 112, KINETICS_MEAN, KINETICS_STD, dtype)


def val_vid(dtype=torch.bfloat16):        # :211-217
    return ClipPreprocessor(112, 112, KINETICS_MEAN, KINETICS_STD, dtype)


def val_transform(dtype=torch.bfloat16):  # :195-201
    return ClipPreprocessor(230, 224, IMAGENET_MEAN, IMAGENET_STD, dtype)
