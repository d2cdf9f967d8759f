"""Dense inverse path (constant Sigma, GrInt, device-resident) at the per-GPU shares BASELINE's multi-GPU
configurations leave one GPU: C4 sharded 8 ways = 61 contour matrices of N = 800, C5 = 128 matrices of N = 1000,
beside the full-batch figures.  Kernel time of the inverse family from the library's hipEvents."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import random_system
from gaunegf_amd.engine import get_engine
from gaunegf_amd.matTools import formSigma
cases = [(800, 61), (800, 122), (800, 486), (1000, 64), (1000, 128), (1000, 1000), (500, 64), (500, 250), (500, 1000)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
eng = get_engine()
last_n = None
for n, m in cases:
    if n != last_n:
        F, S = random_system(n, seed=1)
        sig = [formSigma(list(range(20)), -0.1j, n, S), formSigma(list(range(n - 20, n)), -0.1j, n, S)]
        eng.set_system(F, S)
        h = eng.sigma_const(sig)
        last_n = n
    eng.set_batch(m)
    dev = torch.device("cuda", eng.device)
    E = torch.complex(torch.linspace(-3, 3, m, dtype=torch.float64), torch.full((m,), 1e-3, dtype=torch.float64)).to(dev)
    w = torch.full((m,), 1.0 / m, dtype=torch.complex128, device=dev)
    out = torch.zeros((n, n), dtype=torch.complex128, device=dev)
    for _ in range(2):
        eng.gr_int_dev(h, m, E.data_ptr(), w.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    reps = 5
    eng.profile(True); eng.profile_reset()
    t = time.perf_counter()
    for _ in range(reps):
        eng.gr_int_dev(h, m, E.data_ptr(), w.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    ims, nl = eng.profile_read("inverse"); eng.profile(False)
    print(f"n={n} m={m}: pass {dt*1e3:.2f} ms, inverse {ims/reps:.2f} ms = {8*n**3*m/(ims/reps*1e-3)/1e12:.1f} TF "
          f"({8*n**3*m/(ims/reps*1e-3)/1e12/78.6:.3f} of peak)", flush=True)
