"""Random sizes through the dense-product paths (G Gamma G^H as a Hermitian product, the transmission's one and a half
products, both zgemm kernels, ragged block edges, odd / even block counts) against the numpy oracle."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tests.helpers import random_system, const_sigma_pair, rel_fro
from gaunegf_amd.integrate import GrLessInt
from gaunegf_amd.surfGTester import surfGTest
from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
from gaunegf_amd.engine import get_engine
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
eng = get_engine()
worst = 0.0
for it in range(16):
    n = int(rng.integers(20, 420)); m = int(rng.integers(1, 9))
    F, S = random_system(n, int(rng.integers(1 << 30)))
    nc = max(2, n // 12)
    inds, s1, s2 = const_sigma_pair(n, S, nc, 0.1)
    dense = bool(rng.integers(0, 2))
    eng.set_gamma_algo(1 if dense else 0)
    g_dev = surfGTest(F, S, inds, -0.1j); g_ref = oracle.ConstSigma(F, S, inds, -0.1j)
    E = np.sort(rng.uniform(-1.5, 1.5, m)) + 0.0j; w = rng.uniform(0.1, 1.0, m) + 0.0j
    for ind in (None, 0, -1):
        r = rel_fro(GrLessInt(F, S, g_dev, E, w, ind), oracle.GrLessInt(F, S, g_ref, E, w, ind)); worst = max(worst, r)
        assert r < 1e-8, (n, m, ind, dense, r)
    T = calculate_transmission(F, S, SigmaCalculator(s1, s2), np.real(E))
    g1, g2 = 1j * (s1 - s1.conj().T), 1j * (s2 - s2.conj().T)
    for k, e in enumerate(np.real(E)):
        G = np.linalg.inv(e * S - F - s1 - s2)
        ref = np.real(np.trace(g1 @ G @ g2 @ G.conj().T))
        assert abs(T[k] - ref) <= 1e-8 * max(1.0, abs(ref)), (n, k, T[k], ref)
    print(f"n={n} m={m} dense={dense}: ok", flush=True)
eng.set_gamma_algo(0)
print("fuzz ok, worst G< rel error", worst)
