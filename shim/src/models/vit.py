"""Drop-in for the reference's ``src/models/vit.py:8-128``: the same names, served by the MI355X build."""
from dvt_amd.models.vit import PreNorm, FeedForward, Attention, Transformer, ViViT  # noqa: F401

__all__ = ['PreNorm', 'FeedForward', 'Attention', 'Transformer', 'ViViT']
