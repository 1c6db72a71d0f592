"""The CPUs this process may actually use, and thread pools sized to them for the duration of a fit.

Inside a container the visible core count (``os.cpu_count()``: 256 on the MI355X boxes) is not the budget: the cgroup's CFS quota
is (``cpu.max`` = 16 CPUs there).  numpy's BLAS and torch's OpenMP pool size themselves by the VISIBLE cores; every small
host-side torch operation of a fit (a 5 x 262144 array converted, summed, copied) then wakes 128 threads that spin after their
region, the process burns its whole quota (5.4-6.4 CPU-seconds in a 0.4 s fit), and the kernel stops EVERY thread of the container
until the next 100 ms period: the ~80 ms "stall" that wandered between the sections of a whole fit in rounds 2 and 3 (device idle,
host threads frozen - DESIGN.md section 7; `profiles/r03w_fit_timing_cpustat.log`: 3-4 of a fit's 4 periods throttled;
`profiles/r03x_fit_timing_omp{16,8,4}.log`: none, and the fit takes 0.18-0.21 s instead of 0.32-0.42).
"""
import contextlib
import math
import os


def _cgroup_quota():
    """CPUs granted by the CFS quota of this process's cgroup (the tightest one on the way to the root), or None."""
    best = None
    # cgroup v2: "0::/path"
    try:
        path = "/"
        for line in open("/proc/self/cgroup").read().splitlines():
            parts = line.split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                path = parts[2]
        p = os.path.normpath("/sys/fs/cgroup/" + path.lstrip("/"))
        while p.startswith("/sys/fs/cgroup"):
            try:
                q, per = open(os.path.join(p, "cpu.max")).read().split()[:2]
                if q != "max" and float(per) > 0:
                    v = float(q) / float(per)
                    best = v if best is None else min(best, v)
            except (OSError, ValueError):
                pass
            if p == "/sys/fs/cgroup":
                break
            p = os.path.dirname(p)
    except OSError:
        pass
    # cgroup v1
    for base in ("/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
        try:
            q = int(open(base + "/cpu.cfs_quota_us").read())
            per = int(open(base + "/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                best = q / per if best is None else min(best, q / per)
        except (OSError, ValueError):
            pass
    return best


def cpu_budget():
    """min(visible cores, affinity mask, cgroup quota), at least 1."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    q = _cgroup_quota()
    if q is not None:
        n = min(n, max(1, int(math.floor(q + 1e-9))))
    return max(1, n)


def fit_threads():
    """Threads per pool during a fit: ``ESPM_CPU_THREADS`` (0 = leave the pools alone), else half the budget, at most 8 - the host
    side of a device fit is small arrays; the budget's other half is the copy threads, the runtime's and the interpreter's."""
    env = os.environ.get("ESPM_CPU_THREADS")
    if env is not None:
        return max(0, int(env))
    return max(1, min(8, cpu_budget() // 2))


_CTL = None


def _controllers():
    """threadpoolctl's controllers of the BLAS / OpenMP libraries in the process, looked up once: a ThreadpoolController walks every
    loaded shared object (several ms with torch in the process) - per fit that is time inside fit_transform."""
    global _CTL
    if _CTL is None:
        from threadpoolctl import ThreadpoolController
        _CTL = list(ThreadpoolController().lib_controllers)
    return _CTL


@contextlib.contextmanager
def limited_thread_pools(n=None):
    """Pools LARGER than ``n`` (torch's intra-op pool; BLAS / OpenMP pools threadpoolctl finds) are cut to ``n`` and restored on
    exit; smaller ones are left as the caller set them."""
    n = fit_threads() if n is None else int(n)
    undo = []
    if n > 0:
        try:
            import torch
            old = torch.get_num_threads()
            if old > n:
                torch.set_num_threads(n)
                undo.append(lambda: torch.set_num_threads(old))
        except Exception:   # noqa: BLE001 - no torch, nothing to cut
            pass
        try:
            for lc in _controllers():
                cur = lc.num_threads
                if cur is not None and cur > n:
                    lc.set_num_threads(n)
                    undo.append(lambda lc=lc, cur=cur: lc.set_num_threads(cur))
        except Exception:   # noqa: BLE001 - threadpoolctl absent or a library it cannot drive
            pass
    try:
        yield n
    finally:
        for f in reversed(undo):
            try:
                f()
            except Exception:   # noqa: BLE001
                pass
