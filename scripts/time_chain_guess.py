"""A provider that has only seen 8 energies evaluates the 2000-point C3 grid: the order predicted from 8 points is a
guess, so the launch runs round robin (negf_api.hip: order_trusted) -- against the same launch forced plain
(negf_set_chain_round_robin(0, 0)) in that guessed order."""
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
E8, w8 = to_dev(Eg[::250]), to_dev(wg[::250])
out = torch.zeros((N, N), dtype=torch.complex128, device=dev)

def evaluate(h, m, E, w):
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.gr_int_dev(h, m, E.data_ptr(), w.data_ptr(), out.data_ptr())
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3

lead0 = surfG(F, S, inds, **kw); h = lead0._negf_lower(eng); evaluate(h, M, E_dev, w_dev)      # allocations (the provider lives as long as its surfG)
for label, q in (("default (guessed order -> round robin)", -1), ("forced plain launch in the guessed order", 0)):
    eng.set_chain_round_robin(q, 0)
    for rep in range(2):
        lead = surfG(F, S, inds, **kw); h = lead._negf_lower(eng)
        evaluate(h, 8, E8, w8)
        print(f"{label}: 2000 points after 8: {evaluate(h, M, E_dev, w_dev):.1f} ms, again: {evaluate(h, M, E_dev, w_dev):.1f} ms", flush=True)
