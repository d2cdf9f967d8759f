"""Where the host time of one FockToP step goes (n200): wall-clock of its sections, by monkeypatching timers around the
callees and timing the inline tail (scfE.py:460-468) separately."""
import sys, os, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.is_available()
import bench
from gaunegf_amd.engine import get_engine
from gaunegf_amd import scfE, density as D
from gaunegf_amd.scfE import NEGFE

name = sys.argv[1] if len(sys.argv) > 1 else "n200"
eng = get_engine()
label, F, S, g, make_ref, ne, Eminf = bench._scf_system(name)
acc = {}
def timed(mod, fn):
    orig = getattr(mod, fn)
    def w(*a, **k):
        t = time.perf_counter()
        try: return orig(*a, **k)
        finally: acc[fn] = acc.get(fn, 0.0) + time.perf_counter() - t
    setattr(mod, fn, w)
for fn in ("calcEmin", "densityReal", "calcFermiMuller", "densityGrid", "densityComplex"):
    timed(scfE, fn) if hasattr(scfE, fn) else None
def new_step():
    n = NEGFE(F, S, g, ne=ne, spin='r', T=300.0, Eminf=Eminf)
    n.setIntegralLimits(tol=1e-4, Emin=None); n.setVoltage(0.1, fermiMethod='muller'); return n
sink = io.StringIO()
with contextlib.redirect_stdout(sink):
    new_step().FockToP()
    for rep in range(3):
        st = new_step(); acc.clear(); torch.cuda.synchronize()
        t0 = time.perf_counter(); st.FockToP(); dt = time.perf_counter() - t0
        # the inline tail, repeated outside
        t1 = time.perf_counter()
        Dd, V = np.linalg.eigh(st.X @ F @ st.X); t2 = time.perf_counter()
        Xi = np.linalg.inv(st.X); t3 = time.perf_counter()
        ps = V.conj().T @ (Xi @ st.P @ Xi) @ V; t4 = time.perf_counter()
        tr = np.trace(S @ st.P).real; t5 = time.perf_counter()
        sys.stderr.write(f"{name} step {dt*1e3:.1f} ms: " + ", ".join(f"{k} {v*1e3:.1f}" for k, v in acc.items()) +
                         f" | tail alone: eigh(XFX) {(t2-t1)*1e3:.1f}, inv(X) {(t3-t2)*1e3:.1f}, products {(t4-t3)*1e3:.1f}, trace(SP) {(t5-t4)*1e3:.1f}; X dtype {st.X.dtype}\n")
