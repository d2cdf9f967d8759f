"""Chain kernel at full occupancy with jobs of EQUAL length (force_iters): 384 energies x 2 contacts = 768 workgroups =
one per resident slot (3 per CU), no tail: kernel time / sweeps = time per sweep and slot; compare with the
free-running C3 grid (scripts/time_chain.py), where job lengths vary from ~100 to 2000 sweeps."""
import sys, time, numpy as np
import os; os.environ.setdefault("NEGF_CHAIN_CACHE", "0")      # time the fixed point, not the g(E) cache
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_configs import _c3_system
from gaunegf_amd.engine import get_engine
F, S, g, ref = _c3_system()
eng = get_engine()
for ne, iters in ((384, 600), (768, 600), (1536, 300)):
    E = np.linspace(-2, 2, ne)
    g.force_iters = iters
    g._version += 1
    g.sigma_batch(E[:8])
    eng.profile(True); eng.profile_reset()
    sig, it, cv = g.sigma_batch(E)
    ms, n = eng.profile_read("chain1d"); eng.profile(False)
    slots = 768
    rounds = -(-2 * ne // slots)
    print(f"{ne} energies x 2 contacts x {iters} sweeps: kernel {ms:.1f} ms -> {ms * 1e3 / (iters * rounds):.1f} us per sweep and slot "
          f"({rounds} round(s) of 768 workgroups), {24*50**3*it.sum()/ms/1e9:.2f} TFLOP/s algorithmic")
