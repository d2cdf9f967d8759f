"""First against repeated evaluation of the C3 grid (2000 energies x 2 leads of n_c = 50): a FRESH provider has no sweep
counts to predict an order from, so its first launch runs in launch order -- unless the launch runs round robin
(NEGF_CHAIN_RR, negf_set_chain_round_robin), which needs no order.  GrInt through the device-pointer entry point,
workspace allocated beforehand, g(E) cache off."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from gaunegf_amd.engine import Engine
from gaunegf_amd.surfG1D import surfG

N, NC, M = 500, 50, 2000
F, S, inds, kw = bench.c3_system(N, NC, 1e-4)
Eg, wg = bench.legendre_grid(M, -2.0, 2.0)
torch.cuda.init()
eng = Engine(0)
stream = torch.cuda.current_stream(); eng.set_stream(stream.cuda_stream)
eng.set_system(F, S); eng.set_chain_cache(0)
dev = torch.device("cuda", 0)
to_dev = lambda a: torch.view_as_complex(torch.from_numpy(np.ascontiguousarray(a, dtype=np.complex128).view(np.float64).reshape(-1, 2).copy())).to(dev)
E_dev, w_dev = to_dev(Eg), to_dev(wg)
out = torch.zeros((N, N), dtype=torch.complex128, device=dev)

def evaluate(h):
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.gr_int_dev(h, M, E_dev.data_ptr(), w_dev.data_ptr(), out.data_ptr())
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3

lead = surfG(F, S, inds, **kw); h = lead._negf_lower(eng); evaluate(h)       # allocations
ref = out.clone()
for rep in range(3):
    lead = surfG(F, S, inds, **kw); lead._engine_override = eng if hasattr(lead, "_engine_override") else None
    h = lead._negf_lower(eng)
    t1 = evaluate(h); ok1 = bool(torch.equal(out, ref))
    t2 = evaluate(h); t3 = evaluate(h)
    print(f"fresh provider: first {t1:.1f} ms  second {t2:.1f} ms  third {t3:.1f} ms   identical to the reference run: {ok1 and bool(torch.equal(out, ref))}", flush=True)
