"""Constants of the multiplicative-update path (espm/conf.py:55-59)."""
log_shift = 1e-14
dicotomy_tol = 1e-5
seed_max = 4294967295
sigmaL = 8
maxit_dichotomy = 100
