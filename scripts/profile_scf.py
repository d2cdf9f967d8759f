"""Host-side profile (cProfile) of one NEGFE.FockToP step of bench.py --config scf: where the wall time that is NOT
kernel time goes.  usage: python scripts/profile_scf.py n60 n200 n800"""
import sys, os, time, cProfile, pstats, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.is_available()
import bench
from gaunegf_amd.engine import get_engine
from gaunegf_amd.scfE import NEGFE

eng = get_engine()
for name in (sys.argv[1:] or ["n60"]):
    label, F, S, g, make_ref, ne, Eminf = bench._scf_system(name)

    def new_step():
        n = NEGFE(F, S, g, ne=ne, spin='r', T=300.0, Eminf=Eminf)
        n.setIntegralLimits(tol=1e-4, Emin=None)
        n.setVoltage(0.1, fermiMethod='muller')
        return n
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        new_step().FockToP()
        st = new_step()
        torch.cuda.synchronize()
        pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); st.FockToP(); pr.disable(); dt = time.perf_counter() - t0
    print(f"==== {name}: one step {dt*1e3:.1f} ms under cProfile", flush=True)
    for key in ("cumulative", "tottime"):
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30)
        print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:5500], flush=True)
