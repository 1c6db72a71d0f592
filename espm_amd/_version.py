"""The one version string of the package (pyproject.toml reads it; espm_amd.__version__ re-exports it)."""
__version__ = "0.3.0"
