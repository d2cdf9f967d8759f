"""Cost of creating and freeing a CONST self-energy provider (device allocations, uploads, frees) with a few GB
of other allocations alive: what a per-call provider used to add to calculate_transmission / calculate_dos."""
import sys, time, numpy as np
import torch
torch.zeros(1, device='cuda')
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import random_system
from gaunegf_amd.engine import get_engine
from gaunegf_amd.matTools import formSigma
n=200
F,S=random_system(n,2)
eng=get_engine(); eng.set_system(F,S)
sig=[formSigma(list(range(20)), -0.1j, n, S), formSigma(list(range(n-20,n)), -0.1j, n, S)]
big = torch.empty(int(3e9)//8, dtype=torch.float64, device='cuda')   # some live allocations, as after GrInt
for k in range(6):
    t0=time.perf_counter(); h=eng.sigma_const(sig); t1=time.perf_counter(); eng.sigma_free(h); t2=time.perf_counter()
    print(f"create {1e3*(t1-t0):.2f} ms  free {1e3*(t2-t1):.2f} ms")
