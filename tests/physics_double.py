"""A small physics-model double for the G-refresh path of the fit loop (espm/estimators/base.py:269-274, :388-392;
interface espm/models/base.py:217-264): a dictionary G that DEPENDS ON W the way an absorption-corrected X-ray model's
does (espm/models/edxs.py:438-500 recomputes G from the current composition) - here a self-absorption factor per
channel, exp(-strength * Abs @ c), with c the normalised mean composition over the first ``m0`` rows of W.  Pure numpy, no
reference code: tests/golden/make_golden.py mixes it into the reference's abstract PhysicalModel to generate fixture F16,
the oracle and the GPU estimator take it as it is (duck-typed)."""
import numpy as np


class AbsorbingModel:
    def __init__(self, G0, Abs, strength, m0):
        self.G0 = np.asarray(G0, dtype=np.float64)
        self.Abs = np.asarray(Abs, dtype=np.float64)
        self.strength = float(strength)
        self.m0 = int(m0)
        self.G = self.G0.copy()
        self.updates = 0           # NMF_update calls that received a W

    def NMF_initialize_W(self, D):
        return np.abs(np.linalg.lstsq(self.G, D, rcond=None)[0])

    def NMF_update(self, W=None):
        if W is not None:
            W = np.asarray(W, dtype=np.float64)
            c = W[:self.m0].sum(axis=1) / W[:self.m0].sum()
            self.G = self.G0 * np.exp(-self.strength * (self.Abs @ c))[:, None]
            self.updates += 1
        return self.G

    def NMF_simplex(self):
        return np.arange(self.m0)
