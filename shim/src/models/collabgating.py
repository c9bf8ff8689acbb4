"""Drop-in for the reference's ``src/models/collabgating.py:2-87``: the same names, served by the MI355X build."""
from dvt_amd.models.collabgating import CollaborativeGating, GatedEmbeddingUnit, ContextGating  # noqa: F401

__all__ = ['CollaborativeGating', 'GatedEmbeddingUnit', 'ContextGating']
