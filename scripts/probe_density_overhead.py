"""Where does the wall time of densityComplex go beyond its integrals?  Instrumented copy (perf_counter per piece)."""
import sys, os, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.is_available()
import bench
from gaunegf_amd import density as D
from gaunegf_amd.integrate import GrIntSegments, GrInt
label, F, S, g, make_ref, ne, Eminf = bench._scf_system("n60")
T = {}
def tick(k, t0):
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
def run(n):
    for _ in range(n):
        t = time.perf_counter(); hw, mid, rad = D._contour(-30.0, 0.2, 300.0); tick("contour", t)
        lv = D._ant_levels(486)
        t = time.perf_counter()
        segs = []
        for N_, x, w, r in lv:
            ph = np.exp(1j * (np.pi / 2 * (x + 1))); z = mid + rad * ph; wz = np.pi / 2 * w * (1j * rad * ph)
            segs.append((z, wz * D.fermi(z, 0.2, 300.0)))
        tick("grids", t)
        t = time.perf_counter(); out = GrIntSegments(F, S, g, segs); tick("GrIntSegments", t)
        t = time.perf_counter(); P = out[0]
        for k in range(1, len(out)): P = P * 0.333 + out[k]
        tick("combine", t)
        t = time.perf_counter(); print("Complex Contour Integration:"); r = (1 + 0j) * np.imag(P) / np.pi; tick("print+imag", t)
sink = io.StringIO()
with contextlib.redirect_stdout(sink):
    run(3); T.clear(); t0 = time.perf_counter(); run(40); tot = time.perf_counter() - t0
print({k: round(v / 40 * 1e3, 3) for k, v in T.items()}, "total per call %.3f ms" % (tot / 40 * 1e3))
# the real function
with contextlib.redirect_stdout(sink):
    D.densityComplex(F, S, g, -30.0, 0.2, tol=1e-4, T=300.0)
    t0 = time.perf_counter()
    for _ in range(40): D.densityComplex(F, S, g, -30.0, 0.2, tol=1e-4, T=300.0)
    tot = time.perf_counter() - t0
print("densityComplex itself: %.3f ms per call" % (tot / 40 * 1e3))
from gaunegf_amd.engine import get_engine
eng = get_engine(); eng.set_system(F, S); h = g._negf_lower(eng)
segs = [(np.linspace(-1, 1, k) + 0.1j, np.ones(k)) for k in (2, 4, 12, 36, 108, 324)]
for _ in range(3): eng.gr_int_seg(h, segs)
t0 = time.perf_counter()
for _ in range(40): eng.gr_int_seg(h, segs)
print("engine.gr_int_seg (486 points, 6 segments): %.3f ms per call" % ((time.perf_counter() - t0) / 40 * 1e3))
