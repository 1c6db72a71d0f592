"""Thread pools of a fit are sized to the CPUs the process may use (espm_amd/_cpu_budget.py), and restored afterwards."""
import os

import pytest
import torch

# (by path: importing the package loads libespm_mu.so, which this module does not need)
import importlib.util
_spec = importlib.util.spec_from_file_location("_espm_cpu_budget_t", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "espm_amd", "_cpu_budget.py"))
cb = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cb)


def test_budget_is_positive_and_within_the_visible_cores():
    n = cb.cpu_budget()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert 1 <= cb.fit_threads() <= max(1, n)


def test_cgroup_quota_is_read(tmp_path, monkeypatch):
    q = cb._cgroup_quota()
    assert q is None or q > 0


def test_pools_are_cut_and_restored(monkeypatch):
    before = torch.get_num_threads()
    if before < 2:
        pytest.skip("one thread: nothing to cut")
    with cb.limited_thread_pools(1) as n:
        assert n == 1 and torch.get_num_threads() == 1
        try:
            from threadpoolctl import threadpool_info
            assert all((i.get("num_threads") or 1) <= 1 for i in threadpool_info())
        except ImportError:
            pass
    assert torch.get_num_threads() == before


def test_env_override_leaves_the_pools_alone(monkeypatch):
    monkeypatch.setenv("ESPM_CPU_THREADS", "0")
    before = torch.get_num_threads()
    with cb.limited_thread_pools() as n:
        assert n == 0 and torch.get_num_threads() == before


def test_larger_limit_does_not_raise_a_pool():
    before = torch.get_num_threads()
    with cb.limited_thread_pools(before + 7):
        assert torch.get_num_threads() == before
