"""Time the 1-D chain self-energy kernel (C3 contact: n_c = 50, eta = 1e-4) on a few energies."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from scripts.bench_configs import _c3_system
F, S, g, ref = _c3_system()
E = np.linspace(-2, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 384)
g.sigma_batch(E[:8])
t0 = time.perf_counter(); sig, it, cv = g.sigma_batch(E); t = time.perf_counter() - t0
print(f"{len(E)} energies x 2 contacts: {t*1e3:.1f} ms, mean sweeps {it.mean():.0f}, {it.sum()/t/1e6:.2f} M sweeps/s")
