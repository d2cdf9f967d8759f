import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import random_system
from gaunegf_amd.engine import get_engine
from gaunegf_amd.matTools import formSigma
eng = get_engine()
for n in (500, 400, 448, 400):
    m = 1000
    F, S = random_system(n, 1)
    sig = [formSigma(list(range(20)), -0.1j, n, S), formSigma(list(range(n-20, n)), -0.1j, n, S)]
    E = np.linspace(-3, 3, m) + 1e-3j; w = np.ones(m, dtype=complex)/m
    t0=time.perf_counter(); eng.set_system(F, S); t1=time.perf_counter(); h = eng.sigma_const(sig); t2=time.perf_counter()
    eng.gr_int(h, E, w)
    eng.profile(True); eng.profile_reset()
    t3=time.perf_counter(); eng.gr_int(h, E, w); t4=time.perf_counter()
    inv = eng.profile_read("inverse"); eng.profile(False)
    print(n, f"set_system {1e3*(t1-t0):.1f} ms, sigma_const {1e3*(t2-t1):.1f} ms, gr_int wall {1e3*(t4-t3):.1f} ms, inverse (events) {inv[0]:.1f} ms")
    eng.sigma_free(h)

# front-end (gaunegf_amd.integrate.GrInt) against the raw engine call, same system
import gaunegf_amd.integrate as gi
from gaunegf_amd.surfGTester import surfGTest
for n in (400, 500):
    m = 1000
    F, S = random_system(n, 1)
    g = surfGTest(F, S, [list(range(20)), list(range(n - 20, n))], -0.1j)
    E = np.linspace(-3, 3, m) + 1e-3j; w = np.ones(m, dtype=complex) / m
    gi.GrInt(F, S, g, E, w)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); gi.GrInt(F, S, g, E, w); ts.append(1e3 * (time.perf_counter() - t0))
    print(n, "GrInt front-end wall ms:", [round(t, 1) for t in ts])
