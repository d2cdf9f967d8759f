"""Random round-robin configurations of the chain kernel (lead sizes, grids, quanta, slot counts, free-running and fixed
trip counts) against the plain launch of the same grid: Sigma, sweep counts and flags must be identical bit for bit.
usage: python scripts/fuzz_chain_rr.py [seed] [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import chain_lead, random_system
from gaunegf_amd.engine import get_engine
from gaunegf_amd.surfG1D import surfG

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
eng = get_engine()
eng.set_chain_cache(0)
t_end = time.time() + budget
cases = bad = 0
while time.time() < t_end:
    ncL, ncR = (int(x) for x in rng.integers(2, 65, size=2))
    if rng.random() < 0.5:
        ncR = ncL
    N = ncL + ncR + int(rng.integers(1, 12))
    eta = float(10.0 ** rng.uniform(-3.5, -2.0))
    M = int(rng.integers(3, 40))
    F, S = random_system(N, int(rng.integers(1 << 30)))
    aL = chain_lead(ncL, int(rng.integers(1 << 30))); aR = chain_lead(ncR, int(rng.integers(1 << 30)))
    kw = dict(taus=[aL[2].copy(), aR[2].copy()], staus=[aL[3].copy(), aR[3].copy()], alphas=[aL[0], aR[0]],
              aOverlaps=[aL[1], aR[1]], betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=eta)
    inds = [list(range(ncL)), list(range(N - ncR, N))]
    E = np.sort(rng.uniform(-2.0, 2.0, M)) + 0j
    if rng.random() < 0.3:
        E[int(rng.integers(M))] += 1j * rng.uniform(0.05, 0.5)
    force = int(rng.integers(1, 60)) if rng.random() < 0.3 else None
    def run(q, s):
        eng.set_chain_round_robin(q, s)
        g = surfG(F, S, inds, **kw)
        if force is not None:
            g.force_iters = force
        return g.sigma_batch(E)
    ref = run(0, 0)
    for _ in range(3):
        q = int(rng.choice([1, 2, 3, 5, 8, 13, 40, 100, 333])); s = int(rng.integers(1, max(2, 2 * M)))
        got = run(q, s)
        ok = all(np.array_equal(a, b) for a, b in zip(ref, got))
        cases += 1
        if not ok:
            bad += 1
            print(f"MISMATCH nc=({ncL},{ncR}) N={N} eta={eta:.2e} M={M} force={force} quantum={q} slots={s}", flush=True)
eng.set_chain_round_robin(-1, 0)
print(f"seed {seed}: {cases} round-robin launches against the plain launch, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
