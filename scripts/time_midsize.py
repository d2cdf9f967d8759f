import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from tests.helpers import random_system
from gaunegf_amd.engine import get_engine
from gaunegf_amd.surfGTester import surfGTest
import gaunegf_amd.integrate as gi
sizes = [int(a) for a in sys.argv[1:]] or [300, 400, 500]
for n, m in [(k, 1000) for k in sizes]:
    F, S = random_system(n, seed=1)
    g = surfGTest(F, S, [list(range(0, 20)), list(range(n-20, n))], -0.1j)
    E = np.linspace(-3, 3, m) + 1e-3j; w = np.ones(m, dtype=complex)/m
    gi.GrInt(F, S, g, E, w)   # warm-up with the full grid: workspace allocation stays out of the timing
    t=time.perf_counter(); P = gi.GrInt(F, S, g, E, w); dt=time.perf_counter()-t
    print(n, m, f"{dt*1e3:.1f} ms  {8*n**3*m/dt/1e12:.1f} TF")
