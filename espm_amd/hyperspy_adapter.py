"""The hyperspy side of the drop-in (reference: espm/datasets/eds_spim.py:113-139, :484-604,
:606-688, espm/hyperspy_extension.yaml, pyproject.toml:52-53).

The reference is a hyperspy extension: its signal class ``EDSespm`` exposes ``.X`` (the spectrum image as an (n, p)
matrix), ``.shape_2d`` and a ``decomposition(algorithm=est)`` that hands the unfolded data - (pixels, channels), hyperspy's
layout - to ``est.fit_transform`` (an estimator built with ``hspy_comp=True``), takes the loadings it returns and the
factors from ``est.components_``, and keeps the estimator as ``learning_results.decomposition_algorithm`` (from which the
reporting methods read ``W_``, ``G_``, ``H_`` after checking ``isinstance(..., NMFEstimator)``).

What is here:

* ``SpectrumImage`` - that contract without hyperspy: a light signal class over a (ny, nx, n) data cube with ``X``,
  ``shape_2d``, ``decomposition``, ``learning_results``, ``get_decomposition_loadings / _factors``.  It is what the tests
  drive (hyperspy is not installable in the build image), and it is usable on its own for a cube that did not come from
  hyperspy.  The (pixels, channels) view it hands over is the cube's own memory: with ``hspy_comp=True`` the estimator
  uploads it as it is (pixel-major is the engine's native ingest layout) - no host transpose, no device transpose.
* ``register()`` - where espm itself is installed, registers this package's ``NMFEstimator`` as a virtual subclass of the
  reference's abstract base (``abc.ABC.register``), so that ``EDSespm.plot_1D_results`` / ``concentration_report``
  (eds_spim.py:607, :639) accept a decomposition made with it.  Called by ``decompose`` below and harmless elsewhere.
* ``decompose(signal, est)`` - ``signal.decomposition(algorithm=est)`` on a real hyperspy signal (or a ``SpectrumImage``)
  with the checks the reference leaves to the user: ``hspy_comp`` must be on, ``shape_2d`` is taken from the signal when the
  estimator has none (the Laplacian needs the image grid, base.py:286-291).
* ``hyperspy_extension.yaml`` (next to this file) and the ``hyperspy.extensions`` entry point in ``pyproject.toml`` declare
  the signal type ``EDS_espm_amd`` -> ``EDSespmAMD`` below, defined only when hyperspy imports.
"""
from __future__ import annotations

import numpy as np


class LearningResults:
    """The attributes of hyperspy's ``LearningResults`` the reference reads (eds_spim.py:607-612, :639-644)."""

    def __init__(self):
        self.decomposition_algorithm = None
        self.factors = None          # (n, k)
        self.loadings = None         # (p, k)
        self.output_dimension = None
        self.navigation_mask = None
        self.signal_mask = None


class SpectrumImage:
    """A spectrum image (ny, nx, n) with the decomposition contract of hyperspy's ``Signal1D`` / the reference's ``EDSespm``.

    ``data`` is kept as given (no copy); ``X`` and the matrix handed to the estimator are views of it when it is
    C-contiguous."""

    def __init__(self, data):
        data = np.asarray(data)
        if data.ndim != 3:
            raise ValueError("a spectrum image is (rows, columns, channels)")
        self.data = data
        self.learning_results = LearningResults()

    @property
    def shape_2d(self):
        """(rows, columns) of the image, eds_spim.py:113-120."""
        return self.data.shape[0], self.data.shape[1]

    @property
    def X(self):
        """The data as (channels, pixels), eds_spim.py:122-131 (a view for a C-contiguous cube)."""
        ny, nx, n = self.data.shape
        return self.data.reshape((ny * nx, n)).T

    def unfolded(self):
        """(pixels, channels): what hyperspy's decomposition hands to a custom algorithm."""
        ny, nx, n = self.data.shape
        return self.data.reshape((ny * nx, n))

    def decomposition(self, algorithm, output_dimension=None, return_info=False, **kwargs):
        """hyperspy's ``decomposition(algorithm=<object>)`` for a custom estimator: ``fit_transform(data (p, n))`` ->
        loadings (p, k), ``components_`` (k, n) -> factors (n, k); the estimator stays in ``learning_results``."""
        if not hasattr(algorithm, "fit_transform"):
            raise ValueError("algorithm must implement fit_transform() (scikit-learn style)")
        if kwargs:
            raise TypeError(f"unsupported decomposition arguments for a custom algorithm: {sorted(kwargs)}")
        loadings = algorithm.fit_transform(self.unfolded())
        factors = np.asarray(algorithm.components_).T
        lr = self.learning_results
        lr.decomposition_algorithm = algorithm
        lr.loadings, lr.factors = np.asarray(loadings), factors
        lr.output_dimension = factors.shape[1] if output_dimension is None else output_dimension
        if return_info:
            return algorithm

    def get_decomposition_loadings(self):
        """(k, ny, nx) maps."""
        ny, nx = self.shape_2d
        return self.learning_results.loadings.T.reshape((-1, ny, nx))

    def get_decomposition_factors(self):
        """(k, n) spectra."""
        return self.learning_results.factors.T


def register():
    """``espm.estimators.NMFEstimator.register(espm_amd.estimators.NMFEstimator)`` where espm is importable: the reference's
    ``isinstance(learning_results.decomposition_algorithm, NMFEstimator)`` gates then accept this package's estimators.
    Returns True when registered."""
    from espm_amd.estimators import NMFEstimator
    try:
        from espm.estimators import NMFEstimator as RefBase
    except Exception:
        return False
    RefBase.register(NMFEstimator)
    return True


def decompose(signal, est, **kwargs):
    """``signal.decomposition(algorithm=est)`` with the estimator set up for hyperspy's calling convention."""
    if not getattr(est, "hspy_comp", False):
        raise ValueError("hyperspy hands (pixels, channels) to the estimator: build it with hspy_comp=True "
                         "(espm/estimators/base.py:249-259 only warns)")
    if getattr(est, "shape_2d", None) is None and hasattr(signal, "shape_2d"):
        est.shape_2d = tuple(int(v) for v in signal.shape_2d)
    register()
    signal.decomposition(algorithm=est, **kwargs)
    return signal.learning_results


try:  # the real signal class, where hyperspy is installed (hyperspy_extension.yaml names it)
    import hyperspy.api as _hs

    class EDSespmAMD(_hs.signals.Signal1D):
        """hyperspy signal with the reference's ``X`` / ``shape_2d`` accessors (eds_spim.py:113-131); decompositions go
        through hyperspy's own ``decomposition`` with an ``espm_amd`` estimator as the ``algorithm`` object."""
        _signal_type = "EDS_espm_amd"

        @property
        def shape_2d(self):
            return self.axes_manager[1].size, self.axes_manager[0].size

        @property
        def X(self):
            shape = self.axes_manager[1].size, self.axes_manager[0].size, self.axes_manager[2].size
            return self.data.reshape((shape[0] * shape[1], shape[2])).T
except Exception:  # pragma: no cover - hyperspy is absent in the build image
    EDSespmAMD = None
