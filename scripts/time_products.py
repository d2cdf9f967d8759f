"""Time the dense product paths (GrLessInt, transmission) at a few sizes with the library's own
hipEvent profile; prints per-family milliseconds."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from tests.helpers import random_system
from gaunegf_amd.engine import get_engine
from gaunegf_amd.matTools import formSigma
eng = get_engine()
for n, m in ((200, 1000), (500, 400), (1000, 128)):
    F, S = random_system(n, 1)
    nc = n // 10
    sig = [formSigma(list(range(nc)), -0.1j, n, S), formSigma(list(range(n - nc, n)), -0.1j, n, S)]
    eng.set_system(F, S)
    h = eng.sigma_const(sig)
    E = np.linspace(-3, 3, m) + 0j; w = np.ones(m, dtype=complex) / m
    eng.gless_int(h, -1, E, w); eng.transmission(h, 0, 1, E)
    eng.profile(True); eng.profile_reset()
    t0 = time.perf_counter(); eng.gless_int(h, -1, E, w); t1 = time.perf_counter(); eng.transmission(h, 0, 1, E); t2 = time.perf_counter()
    fam = {f: eng.profile_read(f) for f in ("inverse", "zgemm", "gamma", "accumulate", "trace", "assemble")}
    eng.profile(False)
    zg_ms, zg_n = fam["zgemm"]
    print(f"n={n} m={m}: gless {1e3*(t1-t0):.2f} ms, transmission {1e3*(t2-t1):.2f} ms; " +
          ", ".join(f"{k} {v[0]:.2f} ms/{v[1]}" for k, v in fam.items()) +
          f"; zgemm {4 * 8.0 * n**3 * m / (zg_ms * 1e-3) / 1e12:.1f} TF")
    eng.sigma_free(h)
