"""espm_amd - MI355X-native SmoothNMF multiplicative-update path of adriente/espm.

Drop-in surface (same names as the reference): ``espm_amd.estimators.SmoothNMF`` /
``NMFEstimator``, ``espm_amd.estimators.updates.multiplicative_step_h`` / ``_w`` /
``initialize_algorithms``, ``espm_amd.estimators.dicotomy.dichotomy_simplex``,
``espm_amd.measures``, ``espm_amd.utils.create_laplacian_matrix``.
The arithmetic runs in ``lib/libespm_mu.so`` (hand-written HIP for gfx950); importing the
package without that library fails loudly.
"""
from . import _lib  # noqa: F401  (loads libespm_mu.so or raises)

from ._version import __version__  # noqa: E402,F401
