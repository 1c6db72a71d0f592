"""CPU oracle for the SmoothNMF multiplicative-update path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``espm_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker / the reported CPU baseline - never as the thing shipped.

``oracle.mu_oracle`` is a from-scratch numpy (fp64) restatement of the reference's
algorithm for this path (adriente/espm v1.1.3: espm/estimators/updates.py, dicotomy.py,
smooth_nmf.py, base.py, espm/measures.py, espm/utils.py).  Parity is PINNED: every function
is checked against golden vectors produced by importing the unmodified reference in the
build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
checked by ``tests/test_oracle_golden.py``).
"""
