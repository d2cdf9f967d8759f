"""The two product forms of zgemm_mfma_kernel with BOTH operands batched: G Gamma_b (A B) and X G^H (A B^H) at n = 1000,
256 energies, through GrLessInt over a PRECOMPUTED provider (a coupling matrix per energy).  Run with NEGF_ZGEMM_HERM=0
under a kernel trace: with a constant-Sigma provider (scripts/time_products.py) the first product's B operand is ONE
matrix shared by the whole batch, which is what made the A B form look faster than A B^H (profiles/r04_zgemm_forms.txt)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import random_system
from gaunegf_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eng = get_engine()
eng.set_gamma_algo(1)
F, S = random_system(n, seed=1)
eng.set_system(F, S)
rng = np.random.default_rng(5)
tot = np.zeros((m, n, n), dtype=np.complex128)
tot[:, np.arange(n), np.arange(n)] = -0.1j
gam = (rng.standard_normal((m, 1, n, n)) + 0j) * 0.01
h = eng.sigma_precomputed(tot, gammas=gam)
del tot, gam
dev = torch.device("cuda", eng.device)
E = torch.complex(torch.linspace(-3, 3, m, dtype=torch.float64), torch.full((m,), 1e-3, dtype=torch.float64)).to(dev)
w = torch.full((m,), 1.0 / m, dtype=torch.complex128, device=dev)
out = torch.zeros((n, n), dtype=torch.complex128, device=dev)
for _ in range(2):
    eng.gless_int_dev(h, 0, m, E.data_ptr(), w.data_ptr(), out.data_ptr())
torch.cuda.synchronize()
eng.profile(True); eng.profile_reset()
for _ in range(3):
    eng.gless_int_dev(h, 0, m, E.data_ptr(), w.data_ptr(), out.data_ptr())
torch.cuda.synchronize()
zms, nl = eng.profile_read("zgemm"); eng.profile(False)
print(f"n={n} m={m}: two products {zms/3:.2f} ms ({nl} launches)", flush=True)
