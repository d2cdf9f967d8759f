"""Random sizes / batch sizes through the blocked inverses (GrBatch), each result checked by its residual G A = I:
edge cases of the window pairs, sub-panel pairs, narrow last windows, the eight-wave / lean update selection."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import random_system
from gaunegf_amd.integrate import GrBatch
from gaunegf_amd.surfGTester import surfGTest
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = [(int(rng.integers(257, 700)), int(rng.integers(1, 40))) for _ in range(14)]
cases += [(int(rng.integers(257, 520)), int(rng.integers(100, 420))) for _ in range(4)]
cases += [(int(rng.integers(1025, 1400)), int(rng.integers(1, 6))) for _ in range(3)]
cases += [(int(rng.integers(2049, 2300)), 2), (int(rng.integers(4097, 4300)), 1), (int(rng.integers(33, 256)), 50)]
worst = 0.0
for n, m in cases:
    F, S = random_system(n, int(rng.integers(1 << 30)))
    nc = max(2, n // 20)
    g = surfGTest(F, S, [list(range(nc)), list(range(n - nc, n))], -0.1j)
    E = np.linspace(-2.0, 2.0, m) + 0.02j
    G = GrBatch(F, S, g, E)
    sig = g.sigmaTot(0.0)
    res = 0.0
    for k in sorted(set([0, m // 2, m - 1] + list(rng.integers(0, m, size=min(m, 3))))):
        A = E[k] * S - F - sig
        res = max(res, np.linalg.norm(G[k] @ A - np.eye(n)) / np.sqrt(n))
    worst = max(worst, res)
    print(f"n={n} m={m}: residual {res:.2e}", flush=True)
    assert res < 1e-9, (n, m, res)
print("fuzz ok, worst residual", worst)
