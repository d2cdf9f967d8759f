"""Where the host time of a per-GPU share of BASELINE C4 / C5 goes, WITHOUT a profiler (cProfile charges its own overhead to
the small wrappers): wall-clock timers around the engine's entry points and the front end's helpers, inclusive, per step.
Usage (GPU box): python scripts/time_share_host.py c5 [steps]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from gaunegf_amd import density as DN, integrate as GI, engine as EN, transport as TR
from gaunegf_amd.engine import get_engine

config = sys.argv[1] if len(sys.argv) > 1 else "c5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = get_engine()
acc = collections.defaultdict(float); cnt = collections.Counter()


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t; cnt[label] += 1
    setattr(obj, name, wrapper)


for name in ("set_system", "gless_int", "transmission", "gr_int_seg", "gr_int", "sigma_const", "_c128_keyed"):
    timed(eng, name, "engine." + name)
timed(GI, "_split_blocks"); timed(GI, "_spin_split"); timed(EN, "fingerprint")
k = 8
if config == "c4":
    from gaunegf_amd.surfGBethe import surfGB
    N = 800
    F, S = bench.random_system(N, 4)
    coords, orbMap, orbTyp = bench._bethe_contacts(N)
    g = surfGB.from_arrays(F, S, [[1, 2, 3], [4, 5, 6]], orbMap, orbTyp, coords, latFile=os.path.join(bench.ROOT, "gaunegf_amd", "data", "Au"), eta=1e-6, fermi=0.0)
    Ec, wc = DN.contour_grid(-8.0, 0.0, 486, 0.0); Er, wr = DN.real_axis_grid(-1e6, -8.0, 256, 0.0)
    Ec, wc, Er, wr = Ec[0::k], wc[0::k], Er[0::k], wr[0::k]
    step = lambda: GI.GrIntSegments(F, S, g, [(Ec, wc), (Er, wr)])
else:
    from gaunegf_amd.matTools import formSigma
    from gaunegf_amd.surfGTester import surfGTest
    N = 1000
    Fa, Sa = bench.random_system(N, 5); Fb, _ = bench.random_system(N, 6)
    Z = np.zeros((N, N))
    F = np.block([[Fa, Z], [Z, Fb]]); S = np.kron(np.eye(2), Sa)
    nc = 30
    left = list(range(nc)); right = list(range(N - nc, N))
    s1 = formSigma(left, -0.1j, N, Sa); s2 = formSigma(right, -0.1j, N, Sa)
    g = surfGTest(F, S, [left + [N + i for i in left], right + [N + i for i in right]], -0.1j)
    sc = TR.SigmaCalculator(s1, s2)
    Eg, wg = DN.bias_window_grid(-0.25, 0.25, 512, 300.0)
    Et = np.real(np.asarray(Eg)).copy()
    Eg, wg, Et = Eg[0::k], wg[0::k], Et[0::k]
    timed(GI, "GrLessInt"); timed(TR, "calculate_transmission")
    step = lambda: (GI.GrLessInt(F, S, g, Eg, wg, -1), TR.calculate_transmission(F, S, sc, Et, spin='u'))
step(); step()
torch.cuda.synchronize()
acc.clear(); cnt.clear()
eng.profile(True); eng.profile_reset()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
kern = sum(eng.profile_read(f)[0] for f in ("inverse", "zgemm", "bethe", "assemble", "accumulate", "gamma", "trace")) / steps
eng.profile(False)
print(f"{config} share: wall {wall:.2f} ms per step, kernel families {kern:.2f} ms")
for name in sorted(acc, key=lambda n: -acc[n]):
    print(f"  {name:34s} {acc[name] / steps * 1e3:8.3f} ms per step  ({cnt[name] / steps:.0f} calls)")
