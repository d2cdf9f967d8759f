"""Small dense host products of the density step (scfE.py FockToP: S @ P, X F X, the Lowdin occupations) against the
BLAS thread count: on a many-core host the default (all cores) is the slowest setting for n = 60 ... 800."""
import time, sys
import numpy as np
from threadpoolctl import threadpool_limits, threadpool_info
print([ (d.get("internal_api"), d.get("num_threads")) for d in threadpool_info()])
for n in (60, 200, 800):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)); B = A.T.copy()
    H = (A + A.conj().T) / 2
    for t in (None, 64, 16, 8, 4, 1):
        ctx = threadpool_limits(limits=t) if t else None
        try:
            for _ in range(2): A @ B
            t0 = time.perf_counter()
            for _ in range(10): A @ B
            mm = (time.perf_counter() - t0) / 10
            t0 = time.perf_counter(); np.linalg.eigh(H); eh = time.perf_counter() - t0
            t0 = time.perf_counter(); np.linalg.inv(A); iv = time.perf_counter() - t0
        finally:
            if ctx is not None: ctx.restore_original_limits()
        print(f"n={n} threads={t or 'default'}: matmul {mm*1e3:.2f} ms  eigh {eh*1e3:.2f} ms  inv {iv*1e3:.2f} ms", flush=True)
