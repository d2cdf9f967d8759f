"""What the predicted job order is worth on a NEW grid: the C3 leads, a 2000-point grid A, then a 1900-point grid B moved by
0.013 eV -- B evaluated by a fresh provider (launch order) against B evaluated after A (order predicted from A)."""
import sys, time, numpy as np
import os; os.environ.setdefault("NEGF_CHAIN_CACHE", "0")      # time the fixed point, not the g(E) cache
os.environ.setdefault("NEGF_CHAIN_RR", "0")                    # ... one workgroup per job: the round-robin launch does not depend on the order
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from gaunegf_amd.surfG1D import surfG
F, S, inds, kw = bench.c3_system(500, 50, 1e-4)
A = np.linspace(-2.0, 2.0, 2000)
B = np.linspace(-2.0, 2.0, 1900) + 0.013
def run(g, E):
    torch.cuda.synchronize(); t = time.perf_counter(); g.sigma_batch(E); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
g1 = surfG(F, S, inds, **kw); g1.sigma_batch(A[:8])            # provider set-up, a tiny first evaluation (prediction from 8 points)
g0 = surfG(F, S, inds, **kw)
print(f"grid B, fresh provider (launch order): {run(g0, B):.0f} ms")
g2 = surfG(F, S, inds, **kw)
print(f"grid A, fresh provider (launch order): {run(g2, A):.0f} ms")
print(f"grid B after grid A (order predicted from A): {run(g2, B):.0f} ms")
print(f"grid B again (its own order): {run(g2, B):.0f} ms")
