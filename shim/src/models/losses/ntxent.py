"""Drop-in for the reference's ``src/models/losses/ntxent.py:5-75``: the same names, served by the MI355X build."""
from dvt_amd.models.losses.ntxent import NT_Xent, ContrastiveLoss  # noqa: F401

__all__ = ['NT_Xent', 'ContrastiveLoss']
