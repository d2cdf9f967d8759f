"""Host profile of the SCF call pattern (bench.py --config scf): cProfile around the timed FockToP steps only, one
table per system.  Usage (GPU box): python scripts/prof_scf_host.py n60,n200 [steps]"""
import contextlib, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaunegf_amd.scfE import NEGFE

names = (sys.argv[1] if len(sys.argv) > 1 else "n60,n200").split(",")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
limits = bench._blas_limits()
ctx = limits(limits=16) if limits else contextlib.nullcontext()
import gc
with ctx:
    for name in names:
        label, F, S, g, make_ref, ne, Eminf = bench._scf_system(name)

        def new_step():
            n = NEGFE(F, S, g, ne=ne, spin='r', T=300.0, Eminf=Eminf)
            n.setIntegralLimits(tol=1e-4, Emin=None)
            n.setVoltage(0.1, fermiMethod='muller')
            return n
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            new_step().FockToP()
            todo = [new_step() for _ in range(steps)]
            gc.collect(); gc.freeze()
            t0 = time.perf_counter()
            for n in todo:
                n.FockToP()
            plain = (time.perf_counter() - t0) / steps
            todo = [new_step() for _ in range(steps)]
            pr = cProfile.Profile(); pr.enable()
            for n in todo:
                n.FockToP()
            pr.disable()
        print(f"== {name}: {plain*1e3:.2f} ms per step unprofiled; table = {steps} steps under cProfile")
        out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28); print(out.getvalue()[out.getvalue().find("ncalls"):])
