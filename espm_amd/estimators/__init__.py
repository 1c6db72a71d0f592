"""NMF estimators with the reference's names (espm/estimators/__init__.py)."""
from espm_amd.estimators.base import NMFEstimator
from espm_amd.estimators.smooth_nmf import SmoothNMF

__all__ = ["NMFEstimator", "SmoothNMF"]
