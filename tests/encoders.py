"""Test-only stand-in for the CNN encoders of FrameTransformer (``vid_encoder=`` / ``img_encoder=``): patchify ->
Linear -> mean over patches -> Linear(d_out), on the product's own operators, so that the token / cross-modal paths
can be checked against the oracle at toy sizes without a 17-layer BatchNorm stack in between.  Its oracle counterpart
is ``oracle_patch_linear_encoder`` below."""
import torch
from torch import nn

from oracle import clip_path as O


class PatchLinearEncoder(nn.Module):
    """Stand-in for the CNN encoders: patchify -> Linear -> mean over
    patches -> Linear(d_out).  Accepts [N, C, H, W] frames or [N, C, T, H, W] chunks."""

    def __init__(self, in_channels=3, patch=16, width=256, d_out=896, compute_dtype=torch.bfloat16):
        super().__init__()
        self.patch = patch
        self.embed = nn.Linear(in_channels * patch * patch, width)
        self.fc = nn.Linear(width, d_out)
        self.compute_dtype = compute_dtype

    def forward(self, x):
        if x.dim() == 5:                                  # [N, C, T, H, W] -> frames
            n, c, t, h, w = x.shape
            x = x.permute(0, 2, 1, 3, 4).reshape(n * t, c, h, w)
        else:
            n, t = x.shape[0], 1
        emb = _F().patch_embed(x, self.embed.weight, self.embed.bias, self.patch, self.compute_dtype)
        tokens = emb.view(n, -1, emb.shape[-1])           # patches of all frames of a chunk
        return _F().linear(_F().mean_rows(tokens), self.fc.weight, self.fc.bias)


def _F():
    from dvt_amd import functional
    return functional


def oracle_patch_linear_encoder(x, P, prefix, patch):
    """CPU counterpart: patchify -> Linear -> mean -> Linear on explicit formulas."""
    if x.dim() == 5:
        n, c, t, h, w = x.shape
        x = x.permute(0, 2, 1, 3, 4).reshape(n * t, c, h, w)
    else:
        n = x.shape[0]
    pt = O.patchify(x[None], patch)[0]                       # [frames, np, pd]
    e = O.linear(pt, P[prefix + "embed.weight"], P[prefix + "embed.bias"])
    e = e.reshape(n, -1, e.shape[-1]).mean(dim=1)
    return O.linear(e, P[prefix + "fc.weight"], P[prefix + "fc.bias"])
