"""Where the wall time of a BASELINE C4 / C5 step goes on the host side: cProfile of one step through the drop-in
front-ends (bench.py --config c4|c5 builds the same systems), next to the kernel families' hipEvent times."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from gaunegf_amd import density as DN
from gaunegf_amd.engine import get_engine
from gaunegf_amd.integrate import GrInt, GrLessInt

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
eng = get_engine()
if cfg == "c4":
    from gaunegf_amd.surfGBethe import surfGB
    N = 800
    F, S = bench.random_system(N, 4)
    coords, orbMap, orbTyp = bench._bethe_contacts(N)
    lat = os.path.join(bench.ROOT, "gaunegf_amd", "data", "Au")
    g = surfGB.from_arrays(F, S, [[1, 2, 3], [4, 5, 6]], orbMap, orbTyp, coords, latFile=lat, eta=1e-6, fermi=0.0)
    Ec, wc = DN.contour_grid(-8.0, 0.0, 486, 0.0)
    Er, wr = DN.real_axis_grid(-1e6, -8.0, 256, 0.0)
    step = lambda: (GrInt(F, S, g, Ec, wc), GrInt(F, S, g, Er, wr))
else:
    from gaunegf_amd.matTools import formSigma
    from gaunegf_amd.surfGTester import surfGTest
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
    N = 1000
    Fa, Sa = bench.random_system(N, 5); Fb, _ = bench.random_system(N, 6)
    Z = np.zeros((N, N))
    F = np.block([[Fa, Z], [Z, Fb]]); S = np.kron(np.eye(2), Sa)
    nc = 30
    left = list(range(nc)); right = list(range(N - nc, N))
    s1 = formSigma(left, -0.1j, N, Sa); s2 = formSigma(right, -0.1j, N, Sa)
    g = surfGTest(F, S, [left + [N + i for i in left], right + [N + i for i in right]], -0.1j)
    sc = SigmaCalculator(s1, s2)
    Eg, wg = DN.bias_window_grid(-0.25, 0.25, 512, 300.0)
    Et = np.real(np.asarray(Eg)).copy()
    step = lambda: (GrLessInt(F, S, g, Eg, wg, -1), calculate_transmission(F, S, sc, Et, spin='u'))
for _ in range(2):
    step()
torch.cuda.synchronize()
fams = ("inverse", "zgemm", "bethe", "assemble", "accumulate", "gamma", "trace")
eng.profile(True); eng.profile_reset()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
prof = {k: round(eng.profile_read(k)[0], 2) for k in fams}; eng.profile(False)
print(f"{cfg}: step wall {dt*1e3:.1f} ms; kernel families (hipEvents) {prof} sum {sum(prof.values()):.1f} ms", flush=True)
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
