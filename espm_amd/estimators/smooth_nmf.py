"""SmoothNMF with the reference's constructor, parameter coercions and attributes
(espm/estimators/smooth_nmf.py:11-475), running its multiplicative updates on the GPU."""
from __future__ import annotations

from copy import deepcopy

import numpy as np

from espm_amd._cpu_budget import limited_thread_pools
from espm_amd.conf import dicotomy_tol, log_shift, sigmaL
from espm_amd.estimators.base import NMFEstimator


class SmoothNMF(NMFEstimator):
    r"""NMF with Laplacian smoothness, log sparsity and simplex constraints:

    .. math:: \min_{W, H \ge \epsilon} D_{GKL}(X \| GWH) + \lambda_L tr(H \Delta H^T) + \mu \sum \log(H + \epsilon_{reg})

    Parameters as in espm/estimators/smooth_nmf.py:46-79.  The default solver ``algo="log_surrogate"`` is
    accelerated, with or without ``linesearch`` (gamma_ adapts to the Laplacian surrogate every iteration,
    smooth_nmf.py:376-381) and with ``true_D`` / ``true_H`` tracking; so are ``algo="bmd"`` (both updates in their Bregman
    variant, smooth_nmf.py:358-372, :416-426) for G = None, ``algo="l2_surrogate"`` (H from the quadratic surrogate
    of the Laplacian term, smooth_nmf.py:311-323; with ``l2=True`` the Frobenius W step and data term, :404-413,
    base.py:197-198) and ``algo="projected_gradient"`` (gamma given or the Lipschitz default, with or without its
    linesearch); what is not built raises ``NotImplementedError`` at fit time.
    """

    loss_names_ = NMFEstimator.loss_names_ + ["log_reg_loss"] + ["Lapl_reg_loss"] + ["gamma"]

    def __init__(self, lambda_L=0.0, linesearch=False, mu=0, epsilon_reg=1, algo="log_surrogate",
                 dicotomy_tol=dicotomy_tol, gamma=None, n_components=2, init=None, tol=1e-4, max_iter=200,
                 random_state=None, verbose=1, debug=False, l2=False, G=None, shape_2d=None, normalize=False,
                 log_shift=log_shift, eval_print=10, true_D=None, true_H=None, fixed_H=None, fixed_W=None,
                 hspy_comp=False, no_stop_criterion=False, simplex_H=False, simplex_W=True):
        super().__init__(n_components=n_components, init=init, tol=tol, max_iter=max_iter, random_state=random_state,
                         verbose=verbose, debug=debug, l2=l2, G=G, shape_2d=shape_2d, normalize=normalize,
                         log_shift=log_shift, eval_print=eval_print, true_D=true_D, true_H=true_H, fixed_H=fixed_H,
                         fixed_W=fixed_W, hspy_comp=hspy_comp, no_stop_criterion=no_stop_criterion,
                         simplex_H=simplex_H, simplex_W=simplex_W)
        self.lambda_L = lambda_L
        self.linesearch = linesearch
        self.mu = mu
        self.epsilon_reg = epsilon_reg
        self.dicotomy_tol = dicotomy_tol
        self.algo = algo
        self.gamma = gamma
        self.check_params()

    def check_params(self):
        """Print-and-coerce parameter checks of the reference (espm/estimators/smooth_nmf.py:145-237)."""
        def coerce(ok, name, what, value, shown=None):
            if not ok:
                print(f"The {what} must be {name}")
                print(f"The {what} is set to {value if shown is None else shown}")
            return ok

        if not coerce(isinstance(self.lambda_L, (int, float)), "a float or int", "regularization parameter lambda_L", 0.0):
            self.lambda_L = 0.0
        if not coerce(isinstance(self.linesearch, bool), "a boolean", "linesearch parameter", False):
            self.linesearch = False
        if not coerce(isinstance(self.mu, (int, float, np.ndarray)), "a float, int or np.ndarray",
                      "regularization parameter mu", 0):
            self.mu = 0
        if not coerce(isinstance(self.epsilon_reg, (int, float)), "a float or int",
                      "regularization parameter epsilon_reg", 1):
            self.epsilon_reg = 1
        if not coerce(isinstance(self.algo, str), "a string", "algorithm parameter", "'log_surrogate'"):
            self.algo = "log_surrogate"
        if not coerce(isinstance(self.simplex_H, bool), "a boolean", "simplex_H parameter", False):
            self.simplex_H = False
        if not coerce(isinstance(self.simplex_W, bool), "a boolean", "simplex_W parameter", True):
            self.simplex_W = True
        if not coerce(isinstance(self.dicotomy_tol, (int, float)), "a float or int", "dicotomy_tol parameter", 1e-3):
            self.dicotomy_tol = 1e-3
        if self.gamma is not None and not coerce(isinstance(self.gamma, (int, float, list)), "a float, int, or list",
                                                 "gamma parameter", None):
            self.gamma = None
        if not coerce(isinstance(self.verbose, (bool, int)), "a boolean or int", "verbose parameter", 1):
            self.verbose = 1
        if not coerce(isinstance(self.debug, bool), "a boolean", "debug parameter", False):
            self.debug = False
        if not coerce(isinstance(self.l2, bool), "a boolean", "l2 parameter", False):
            self.l2 = False
        if not coerce(isinstance(self.n_components, int), "an int", "n_components parameter", 2):
            self.n_components = 2
        # value checks
        if self.algo not in ["l2_surrogate", "log_surrogate", "projected_gradient", "bmd"]:
            print("The algorithm must be 'l2_surrogate', 'log_surrogate', 'bmd' or 'projected_gradient'")
            print("The algorithm is set to 'log_surrogate'")
            self.algo = "log_surrogate"
        if not (self.lambda_L >= 0):
            print("The regularization parameter lambda_L must be non-negative")
            print("The regularization parameter lambda_L is set to 0")
            self.lambda_L = 0
        if not (self.epsilon_reg > 0.0):
            print("The regularization parameter epsilon_reg must be positive")
            print("The regularization parameter epsilon_reg is set to 1")
            self.epsilon_reg = 1.0
        if not np.all(np.array(self.mu) >= 0):
            print("The regularization parameter mu must be non-negative")
            print("The regularization parameter mu is set to 0")
            self.mu = 0
        if self.simplex_H and self.simplex_W:  # smooth_nmf.py:218-222
            print("The simplex constraint must be applied to either W or H or none of them")
            print("The simplex constraint is applied to W and not to H")
            self.simplex_W = True
            self.simplex_H = False
        if self.linesearch:
            if self.l2:
                print("The l2 parameter must be False when using linesearch")
                print("The l2 parameter is set to False")
                self.l2 = False
            if not (self.lambda_L > 0):
                print("The regularization parameter lambda_L must be non-zero when using linesearch")
                print("The regularization parameter lambda_L is set to 1")
                self.lambda_L = 1
        if not (self.algo == "l2_surrogate"):
            if self.l2:
                print("The l2 parameter must be False when using the algorithm " + self.algo)
                print("The l2 parameter is set to False")
                self.l2 = False

    # ---- hooks of the base fit loop ---------------------------------------------------------------------
    def _gamma_value(self):
        # the adapted gamma_ first (smooth_nmf.py:470-473 reports gamma_[0] after the linesearch moved it); the initial
        # Lipschitz bound of the projected gradient only while gamma_ is not set yet
        g = getattr(self, "gamma_", None)
        if g is None:
            if self.algo == "projected_gradient" and self.gamma is None:
                return self._pg_gamma()[0]
            g = sigmaL if self.gamma is None else self.gamma
        return g[0] if isinstance(g, list) else g

    def _engine_kwargs(self):
        return dict(lambda_L=self.lambda_L, mu=self.mu, epsilon_reg=self.epsilon_reg,
                    dicotomy_tol=self.dicotomy_tol, sigmaL=float(self._gamma_value()), bregman=self.algo == "bmd",
                    h_rule={"l2_surrogate": 1, "projected_gradient": 2}.get(self.algo, 0),
                    pg_gamma_w=float(self._pg_gamma()[1]) if self.algo == "projected_gradient" else 0.0,
                    frobenius=bool(self.l2))   # (l2 survives _validate only with algo="l2_surrogate", smooth_nmf.py:223-237)

    def _pg_gamma(self):
        """[gamma_H, gamma_W] of the projected gradient: the user's list, or the Lipschitz bounds at W = H = log_shift
        (smooth_nmf.py:297-306) - astronomically large, so that the iterates do not move, exactly like the reference."""
        if self.gamma is not None:
            return self.gamma
        if getattr(self, "_pg_gamma_cache", None) is None:
            from espm_amd.estimators.updates import estimate_Lipschitz_bound_h, estimate_Lipschitz_bound_w
            G = None if self._identity_G else self.G_
            self._pg_gamma_cache = [float(estimate_Lipschitz_bound_h(self.log_shift, self.X_, G, self.n_components,
                                                                      lambda_L=self.lambda_L, mu=self.mu, epsilon_reg=self.epsilon_reg)),
                                    float(estimate_Lipschitz_bound_w(self.log_shift, self.X_, G, self.n_components))]
        return self._pg_gamma_cache

    def _detailed(self, lkl, reg, lap):
        return [lkl, reg, lap, self._gamma_value()]

    def _begin_fit(self):
        # smooth_nmf.py:290-306: gamma_ is fixed on the first iteration
        self.gamma_ = sigmaL if self.gamma is None else deepcopy(self.gamma)
        if self.algo == "projected_gradient" and self.gamma is None:
            self.gamma_ = list(self._pg_gamma())

    def fit_transform(self, X, y=None, W=None, H=None):
        """Fit the model to X (n, p) and return G W (espm/estimators/smooth_nmf.py:239-282)."""
        if self.algo == "projected_gradient":
            # smooth_nmf.py:297-306, :340-353, :427-447; the default gamma (Lipschitz bounds at log_shift) is formed in _pg_gamma,
            # the linesearch runs in the fit loop (base.py here: pg_linesearch_h / _w of the engine)
            if not (self.gamma is None or (isinstance(self.gamma, list) and len(self.gamma) == 2)):
                raise NotImplementedError("algo='projected_gradient' needs gamma=[gamma_H, gamma_W] (or None) on the GPU path")
            if self.simplex_W:
                raise NotImplementedError("Simplex constraint not implemented for W using the projected gradient method")
        self.gamma_ = None
        self._pg_gamma_cache = None
        # (thread pools sized by the visible cores exhaust a container's CPU quota and get the whole process throttled: _cpu_budget.py)
        with limited_thread_pools():
            return super().fit_transform(X, y=y, W=W, H=H)

    def _iteration(self, W, H):
        """One H update then one W update from host arrays (espm/estimators/smooth_nmf.py:284-455)."""
        if getattr(self, "gamma_", None) is None:
            self.gamma_ = sigmaL if self.gamma is None else deepcopy(self.gamma)
        eng = self._get_engine()
        eng.load_state(W, self._local_H(eng, H))   # (sharded fit: this rank's block of image rows; the whole H comes back)
        eng.st.sigma_l = float(self._gamma_value())
        eng.iterate(1, final_loss=False)
        if self.linesearch:  # smooth_nmf.py:376-381
            self.gamma_ = eng.linesearch_step(self._gamma_value())
        return eng.get_W().astype(W.dtype), self._full_H(eng).astype(H.dtype)

    def loss(self, W, H, average=True, X=None):
        """Regularised loss (espm/estimators/smooth_nmf.py:457-475)."""
        return super().loss(W, H, average=average, X=X)
