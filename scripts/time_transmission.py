import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import random_system
from gaunegf_amd.surfGTester import surfGTest
from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
from gaunegf_amd.integrate import GrInt, GrLessInt
from gaunegf_amd.engine import get_engine
import oracle
N, nc, M = 200, 20, 1000
F, S = random_system(N, 2)
inds = [list(range(nc)), list(range(N - nc, N))]
g = surfGTest(F, S, inds, -0.1j)
E, w = oracle.real_axis_grid(-3.0, 3.0, M, 0.0)
sc = SigmaCalculator(g.sig[0], g.sig[1])
eng = get_engine()
for rep in range(2):
    GrInt(F, S, g, E, w); GrLessInt(F, S, g, E, w, -1)
    for k in range(5):
        eng.profile(True); eng.profile_reset()
        t0 = time.perf_counter(); T = calculate_transmission(F, S, sc, E); dt = time.perf_counter() - t0
        fam = {f: round(eng.profile_read(f)[0], 2) for f in ("inverse", "zgemm", "gamma", "trace", "assemble")}
        eng.profile(False)
        print(f"rep {rep} call {k}: wall {dt*1e3:.2f} ms  gpu {fam} batch {eng.get_batch()}")
