"""Random small systems (n = 1 ... 96) through the single-kernel path: every G(E) checked by its residual G A = I and
against numpy's inverse, GrInt against the explicit sum -- random sizes incl. the tile / wave-column edges, random batch
sizes (more points than resident workgroups too), energies close to eigenvalues of the pencil (ill-conditioned A),
dense host-evaluated Sigma as well as constant contacts.  usage: fuzz_small.py [seed] [cases]"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import MockSigma, random_system
from gaunegf_amd.integrate import GrBatch, GrInt
from gaunegf_amd.surfGTester import surfGTest
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
edges = [1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 81, 95, 96]
worst = 0.0
for case in range(ncases):
    n = int(rng.choice(edges)) if case % 3 == 0 else int(rng.integers(1, 97))
    m = int(rng.choice([1, 2, 3, 7, 40, 300, 1300])) if case % 5 == 0 else int(rng.integers(1, 60))
    F, S = random_system(n, int(rng.integers(1 << 30)))
    if case % 4 == 3:
        base = 0.05 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
        g = MockSigma(base, [np.zeros((n, n), dtype=complex)])
        sig_of = g.sigmaTot
    else:
        nc = max(1, n // 10)
        g = surfGTest(F, S, [list(range(nc)), list(range(n - nc, n))], -0.1j if case % 2 else -1e-5j)
        s0 = g.sigmaTot(0.0); sig_of = lambda e: s0
    ev = np.sort(np.real(np.linalg.eigvals(np.linalg.solve(S, F))))
    E = rng.uniform(-3, 3, m) + 1j * rng.choice([1e-6, 1e-3, 0.1, 1.0], m)
    E[::5] = ev[rng.integers(0, n, size=len(E[::5]))] + 1e-7 * (1 + 1j)          # next to an eigenvalue of the pencil
    w = rng.standard_normal(m) + 1j * rng.standard_normal(m)
    G = GrBatch(F, S, g, E)
    acc = np.zeros((n, n), dtype=complex)
    res = 0.0
    for k in range(m):
        A = E[k] * S - F - sig_of(E[k])
        ref = np.linalg.inv(A)
        # residual scaled by the conditioning: |G A - I| <= c eps cond
        r = np.linalg.norm(G[k] @ A - np.eye(n)) / (np.sqrt(n) * max(1.0, np.linalg.cond(A) * 1e-7))
        d = np.linalg.norm(G[k] - ref) / np.linalg.norm(ref) / max(1.0, np.linalg.cond(A) * 1e-7)
        res = max(res, r, d)
        acc += w[k] * G[k]
    got = GrInt(F, S, g, E, w)
    ri = np.linalg.norm(got - acc) / max(np.linalg.norm(acc), 1e-300)
    worst = max(worst, res, ri)
    assert res < 1e-8 and ri < 1e-10, (case, n, m, res, ri)
    if case % 10 == 0:
        print(f"case {case}: n={n} m={m}: residual {res:.2e}, integral {ri:.2e}", flush=True)
print("fuzz ok,", ncases, "cases, worst", worst)
