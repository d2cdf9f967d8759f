"""Time the dense inverse path (constant Sigma, GrInt, 1000 energies, device-resident) at a few sizes:
kernel time of the inverse family from the library's hipEvents."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import random_system
from gaunegf_amd.engine import get_engine
from gaunegf_amd.matTools import formSigma
sizes = [int(a) for a in sys.argv[1:]] or [300, 400, 500]
eng = get_engine()
for n in sizes:
    m = 1000 if n <= 1000 else 256
    F, S = random_system(n, seed=1)
    sig = [formSigma(list(range(20)), -0.1j, n, S), formSigma(list(range(n - 20, n)), -0.1j, n, S)]
    eng.set_system(F, S)
    h = eng.sigma_const(sig)
    dev = torch.device("cuda", eng.device)
    E = torch.complex(torch.linspace(-3, 3, m, dtype=torch.float64), torch.full((m,), 1e-3, dtype=torch.float64)).to(dev)
    w = torch.full((m,), 1.0 / m, dtype=torch.complex128, device=dev)
    out = torch.zeros((n, n), dtype=torch.complex128, device=dev)
    for _ in range(2):
        eng.gr_int_dev(h, m, E.data_ptr(), w.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    eng.profile(True); eng.profile_reset()
    t = time.perf_counter()
    for _ in range(3):
        eng.gr_int_dev(h, m, E.data_ptr(), w.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    ims, nl = eng.profile_read("inverse"); eng.profile(False)
    print(f"n={n} m={m}: pass {dt*1e3:.2f} ms, inverse {ims/3:.2f} ms = {8*n**3*m/(ims/3*1e-3)/1e12:.1f} TF "
          f"({8*n**3*m/(ims/3*1e-3)/1e12/78.6:.3f} of peak)", flush=True)
    eng.sigma_free(h)
