"""Drop-in for the reference's ``src/models/transformer.py:10-175``: the same names, served by the MI355X build."""
from dvt_amd.models.transformer import PositionalEncoding, SimpleTransformer  # noqa: F401

__all__ = ['PositionalEncoding', 'SimpleTransformer']
