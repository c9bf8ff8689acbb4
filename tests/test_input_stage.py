"""the on-device input stage (Resize + CenterCrop + ToTensor + Normalize of
MMX_Light_dl.py:203-217).  CPU: the numpy oracle against fixtures written through Pillow itself.
GPU: the HIP kernels against the fixtures and the oracle, bit-exact in fp32."""
import numpy as np
import pytest
import torch

from oracle import input_stage as I
from tests.util import golden

CASES = ["down_wide", "down_tall", "up", "train_vid", "identity"]


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_pillow_fixture(tag):
    g = golden("input_stage.npz")
    resize, crop = (int(v) for v in g[f"{tag}:cfg"])
    got = I.preprocess_frames(g[f"{tag}:frames"], resize, crop, g["mean"], g["std"])
    assert got.dtype == np.float32 and np.array_equal(got, g[f"{tag}:out"])


def test_oracle_resize_against_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for (H0, W0, S) in ((90, 160, 40), (161, 90, 47), (31, 53, 64), (240, 426, 120)):
        img = rng.integers(0, 256, (H0, W0, 3), dtype=np.uint8)
        h, w = I.resized_hw(H0, W0, S)
        ref = np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
        assert np.array_equal(I.resize_bilinear_u8(img, h, w), ref)


def test_oracle_rejects_crop_larger_than_frame():
    with pytest.raises(ValueError):
        I.preprocess_frames(np.zeros((1, 20, 30, 3), np.uint8), 16, 24, (0, 0, 0), (1, 1, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_input_stage_bit_exact(device, tag):
    from dvt_amd import ops
    g = golden("input_stage.npz")
    resize, crop = (int(v) for v in g[f"{tag}:cfg"])
    frames = torch.from_numpy(g[f"{tag}:frames"]).cuda()
    out = ops.frames_preprocess(frames, resize, crop, g["mean"], g["std"], torch.float32)
    assert torch.equal(out.cpu(), torch.from_numpy(g[f"{tag}:out"]))
    out16 = ops.frames_preprocess(frames, resize, crop, g["mean"], g["std"], torch.bfloat16)
    assert torch.equal(out16.cpu(), torch.from_numpy(g[f"{tag}:out"]).to(torch.bfloat16))


@pytest.mark.gpu
def test_hip_input_stage_full_size_clip_and_errors(device):
    """One reference-sized chunk batch: [2, 13, 12] frames of 135x240 -> [2, 13, 12, 3, 112, 112]."""
    from dvt_amd.input_stage import train_vid, ClipPreprocessor
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (2, 13, 12, 135, 240, 3), dtype=np.uint8)
    out = train_vid(torch.float32)(torch.from_numpy(frames).cuda())
    assert out.shape == (2, 13, 12, 3, 112, 112)
    pick = [(0, 0, 0), (1, 12, 11), (0, 7, 5)]
    ref = I.preprocess_frames(np.stack([frames[i] for i in pick]), 120, 112, (0.43216, 0.394666, 0.37645),
                              (0.22803, 0.22145, 0.216989))
    for j, idx in enumerate(pick):
        assert np.array_equal(out[idx].cpu().numpy(), ref[j])
    with pytest.raises(ValueError, match="exceeds"):           # crop larger than the resized frame
        ClipPreprocessor(16, 24)(torch.zeros(1, 20, 30, 3, dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError, match="uint8"):
        ClipPreprocessor(16, 8)(torch.zeros(1, 20, 30, 3, device="cuda"))
