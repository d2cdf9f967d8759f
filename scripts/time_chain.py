"""Time the 1-D chain self-energy kernel (C3 contact: n_c = 50, eta = 1e-4): kernel time from the
library's hipEvents (negf_profile_*), not the host call (which also scatters and downloads Sigma)."""
import sys, time, numpy as np
import os; os.environ.setdefault("NEGF_CHAIN_CACHE", "0")      # time the fixed point, not the g(E) cache
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_configs import _c3_system
from gaunegf_amd.engine import get_engine
F, S, g, ref = _c3_system()
E = np.linspace(-2, 2, int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 384)
eng = get_engine()
g.sigma_batch(E[:8]); g.sigma_batch(E) if "--warm" in sys.argv else None
eng.profile(True); eng.profile_reset()
t0 = time.perf_counter(); sig, it, cv = g.sigma_batch(E); t = time.perf_counter() - t0
ms, n = eng.profile_read("chain1d")
print(f"{len(E)} energies x 2 contacts: kernel {ms:.1f} ms ({n} launch), host call {t*1e3:.1f} ms, mean sweeps {it.mean():.0f}, "
      f"{it.sum()/ms/1e3:.2f} M sweeps/s, {24*50**3*it.sum()/ms/1e9:.2f} TFLOP/s algorithmic")
