"""Drop-in for the reference's ``src/models/TPN.py:2-112``: the same names, served by the MI355X build."""
from dvt_amd.models.TPN import Feature_Pyramid_Mid, Feature_Pyramid_High, Feature_Pyramid_low, TPN, sum_group, Reasoning  # noqa: F401

__all__ = ['Feature_Pyramid_Mid', 'Feature_Pyramid_High', 'Feature_Pyramid_low', 'TPN', 'sum_group', 'Reasoning']
