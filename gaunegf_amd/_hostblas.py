"""Thread count of the HOST BLAS / LAPACK calls of the density step.

The host side of a density step is small dense linear algebra (the Lowdin occupations of scfE.py:460-468, the orbital
energies behind calcEmin, density.py:497-520, the constant-self-energy Fermi estimate): n = 60 ... 1000.  numpy's BLAS
starts one thread per CPU it sees; on a many-core host -- worse, in a container whose CPU quota is a fraction of the CPUs
it sees -- that is a bad setting (MI355X host, 256 CPUs visible, a cgroup quota of 16: ``eigh`` of n = 800 takes 606 ms
with the default 256 threads and 114 ms with 8; n = 200: 88 ms against 5 ms, ``scripts/time_host_blas.py``; spinning BLAS
threads also burn the quota and the whole process is throttled for the rest of the 100-ms period).  The density step
therefore runs its host algebra under a thread limit: min(8, CPUs of the affinity mask, half the CPU quota of the cgroup)
-- one ``FockToP`` step at n = 200: 100 ms without limit, 101 / 85 / 76 / 80 ms with 16 / 8 / 4 / 1 threads; n = 800:
1443 / 1470 / 1481 / 1508 / 1838 ms (``scripts/gpu_r4_bb.sh``; the spread between runs is of the same size).
``NEGF_HOST_BLAS_THREADS`` overrides the number (0: leave numpy alone).  Needs ``threadpoolctl`` (optional: without it
nothing is limited)."""
import contextlib
import os

_limit = None


def _cgroup_cpus():
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except Exception:
        pass
    return None


def host_threads():
    """Threads the host algebra of a density step runs with (None: no limit is applied)."""
    global _limit
    if _limit is None:
        env = os.environ.get("NEGF_HOST_BLAS_THREADS")
        if env is not None:
            _limit = max(int(env), 0)
        else:
            try:
                n = len(os.sched_getaffinity(0))
            except Exception:
                n = os.cpu_count() or 1
            q = _cgroup_cpus()
            _limit = max(1, min(8, n, q // 2 if q else n))
    return _limit or None


_controller = None


def _blas_controller():
    """threadpoolctl's handle on the BLAS / OpenMP pools loaded so far (looking them up walks every shared object of
    the process, ~1 ms: done once, after numpy and scipy.linalg are in)."""
    global _controller
    if _controller is None:
        try:
            import numpy.linalg                                      # noqa: F401
            import scipy.linalg                                      # noqa: F401
            from threadpoolctl import ThreadpoolController
            _controller = ThreadpoolController()
        except Exception:
            _controller = False
    return _controller or None


@contextlib.contextmanager
def limited():
    """``with limited():`` -- host BLAS calls inside run with host_threads() threads."""
    n = host_threads()
    ctl = _blas_controller() if n is not None else None
    if ctl is None:
        yield
        return
    with ctl.limit(limits=n):
        yield


def limited_call(fn):
    """Decorator: the function's host algebra runs under limited()."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        with limited():
            return fn(*a, **k)
    return wrapper
