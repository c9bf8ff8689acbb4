"""Drop-in for the reference's ``src/models/frame_transformer.py:19-366``: the same names, served by the MI355X build."""
from dvt_amd.models.frame_transformer import PositionalEncoding, TransformerBase, ImgResNet, VidResNet, FrameTransformer  # noqa: F401

__all__ = ['PositionalEncoding', 'TransformerBase', 'ImgResNet', 'VidResNet', 'FrameTransformer']
