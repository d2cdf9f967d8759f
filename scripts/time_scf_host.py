"""Wall-clock timeline of one NEGFE.FockToP step (bench.py --config scf systems) WITHOUT a profiler: inclusive timers around the
engine's entry points and the step's host LAPACK calls; the rest is the front end's own Python / numpy bookkeeping.
Usage (GPU box): python scripts/time_scf_host.py n60,n200 [steps]"""
import collections, contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gaunegf_amd import density as DN
from gaunegf_amd.engine import get_engine
from gaunegf_amd.scfE import NEGFE

names = (sys.argv[1] if len(sys.argv) > 1 else "n60,n200").split(",")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = get_engine()
acc = collections.defaultdict(float); cnt = collections.Counter()


def timed(obj, name, label):
    fn = getattr(obj, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t; cnt[label] += 1
    setattr(obj, name, wrapper)


for name in ("set_system", "gr_int", "gr_int_seg", "gr_int_refine", "gless_int", "gless_int_seg", "dos", "sigma_const"):
    timed(eng, name, "engine." + name)
for name in ("eigvals", "eigh", "eig", "inv", "solve"):
    timed(np.linalg, name, "numpy.linalg." + name)
timed(DN, "fermi", "density.fermi"); timed(DN, "_speculation_budget", "density._speculation_budget")
for name in ("densityReal", "densityGrid", "densityComplex", "calcEmin"):
    timed(DN, name, "density." + name)
import gaunegf_amd.scfE as _SC
for name in ("densityReal", "densityGrid", "densityComplex", "calcEmin", "calcFermiMuller", "calcFermiBisect"):
    timed(_SC, name, "scfE->" + name)
limits = bench._blas_limits()
with (limits(limits=16) if limits else contextlib.nullcontext()):
    for name in names:
        label, F, S, g, make_ref, ne, Eminf = bench._scf_system(name)

        def new_step():
            n = NEGFE(F, S, g, ne=ne, spin='r', T=300.0, Eminf=Eminf)
            n.setIntegralLimits(tol=1e-4, Emin=None)
            n.setVoltage(0.1, fermiMethod='muller')
            return n
        with contextlib.redirect_stdout(io.StringIO()):
            new_step().FockToP()
            todo = [new_step() for _ in range(steps)]
            acc.clear(); cnt.clear()
            eng.profile(True); eng.profile_reset()
            t0 = time.perf_counter()
            for n in todo:
                n.FockToP()
            wall = (time.perf_counter() - t0) / steps * 1e3
            kern = sum(eng.profile_read(f)[0] for f in bench.SCF_FAMILIES) / steps
            eng.profile(False)
        tot = sum(v for k, v in acc.items() if k.startswith("engine.") or k.startswith("numpy.")) / steps * 1e3
        print(f"== {name}: wall {wall:.2f} ms per step, kernel families {kern:.2f} ms, engine + LAPACK calls {tot:.2f} ms, other Python {wall - tot:.2f} ms")
        for k in sorted(acc, key=lambda q: -acc[q]):
            print(f"   {k:34s} {acc[k] / steps * 1e3:8.3f} ms per step ({cnt[k] / steps:.0f} calls)")
