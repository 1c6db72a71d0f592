import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# espm_amd/_cpu_budget.py loaded by path: importing the PACKAGE loads libespm_mu.so (and fails loudly without it), which the
# oracle's tests do not need and the build fixtures of the others produce first
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("_espm_cpu_budget", os.path.join(ROOT, "espm_amd", "_cpu_budget.py"))
_cpu_budget = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_cpu_budget)
# (the ranks the multi-process tests start size their pools at import: keep them within the container's CPUs as well)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_v, str(max(1, _cpu_budget.cpu_budget() // 2)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _pools_within_the_cpu_budget():
    """Thread pools of the test process no larger than the CPUs its container grants (espm_amd/_cpu_budget.py): sized by the visible
    cores they exhaust the quota and the whole session is throttled - the oracle's BLAS calls and every CPU-side torch op."""
    with _cpu_budget.limited_thread_pools(_cpu_budget.cpu_budget()):
        yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load
