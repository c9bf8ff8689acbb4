"""CPU restatement of the video-frame input transform.

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Reference: src/dataloaders/mmx/MMX_Light_dl.py:203-217,224-226 --
``transforms.Compose([Resize(S), CenterCrop(C), ToTensor(), Normalize(mean, std)])`` applied to a PIL RGB frame
(S, C = 120, 112 for training clips, 112, 112 for validation; mean/std = Kinetics statistics).

The arithmetic lives in third-party code that is absent from /root/reference:
  * torchvision (unpinned, not installed here): ``Resize(S)`` on a PIL image = ``img.resize((w', h'), BILINEAR)``
    with the shorter side set to S and the longer to ``int(S * long / short)``; ``CenterCrop`` offsets
    ``int(round((h - C) / 2.0))``; ``ToTensor`` = uint8 -> float32 / 255, HWC -> CHW; ``Normalize`` = (t - mean) / std.
  * Pillow (installed: the resize below is pinned against it by tests/golden/input_stage.npz, written by
    tools/gen_golden.py): ``ImagingResample`` 8-bit path -- per output coordinate a triangle filter of support
    ``max(scale, 1)`` centred on ``(x + 0.5) * scale``, weights normalised in double precision, rounded to 22-bit
    fixed point, accumulated from ``1 << 21``, shifted and clipped to uint8; horizontal pass first, then vertical,
    with a uint8 intermediate.
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resized_hw(h: int, w: int, size: int):
    """torchvision ``Resize(int)``: shorter side -> size, longer -> int(size * long / short)."""
    if w <= h:
        return int(size * h / w), size
    return size, int(size * w / h)


def resample_coeffs(in_size: int, out_size: int):
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the bilinear filter (support 1).
    Returns (xmin[out], count[out], coeff[out, ksize] int32)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.zeros(n, np.float64)
        ww = 0.0
        for x in range(n):
            a = (x + lo - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """img uint8 [H, W, C]; resample along ``axis`` (0 = vertical, 1 = horizontal)."""
    in_size = img.shape[axis]
    xmin, cnt, kk = resample_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt[xx]):
            acc += src[xmin[xx] + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """``PIL.Image.resize((out_w, out_h), BILINEAR)`` on uint8 [H, W, 3]: horizontal pass, then vertical."""
    if img.shape[1] != out_w:
        img = _resample_axis(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _resample_axis(img, out_h, 0)
    return img


def crop_offsets(h: int, w: int, crop: int):
    return int(round((h - crop) / 2.0)), int(round((w - crop) / 2.0))


def preprocess_frames(frames: np.ndarray, resize: int, crop: int, mean, std) -> np.ndarray:
    """frames uint8 [F, H0, W0, 3] -> float32 [F, 3, crop, crop] (Resize + CenterCrop + ToTensor + Normalize)."""
    F_, H0, W0, _ = frames.shape
    h, w = resized_hw(H0, W0, resize)
    if h < crop or w < crop:
        raise ValueError("crop larger than the resized frame (torchvision would pad; the reference never does)")
    top, left = crop_offsets(h, w, crop)
    mean32 = np.asarray(mean, np.float32).reshape(3, 1, 1)
    std32 = np.asarray(std, np.float32).reshape(3, 1, 1)
    out = np.empty((F_, 3, crop, crop), np.float32)
    for f in range(F_):
        r = resize_bilinear_u8(frames[f], h, w)[top: top + crop, left: left + crop]
        t = r.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
        out[f] = (t - mean32) / std32
    return out
