"""Latency of SCF-sized calls through the host-pointer API: GrInt of m = 2 ... 324 points at n = 60 / 200, the segmented
call (2 + 4 + 12 + 36 points), GrLessInt; wall time per call (median of many) next to the kernel time inside it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.is_available()
import bench
from gaunegf_amd.engine import get_engine
from gaunegf_amd.integrate import GrInt, GrLessInt, GrIntSegments
from gaunegf_amd.surfGTester import surfGTest
eng = get_engine()
for N in [int(a) for a in sys.argv[1:]] or [60, 200]:
    F, S = bench.random_system(N, 60)
    nc = N // 10
    g = surfGTest(F, S, [list(range(nc)), list(range(N - nc, N))], -0.1j)
    rng = np.random.default_rng(0)
    for m in (2, 12, 54, 108, 324):
        E = rng.uniform(-2, 2, m) + 0.1j; w = np.ones(m)
        for fn, name in ((lambda: GrInt(F, S, g, E, w), "GrInt"), (lambda: GrLessInt(F, S, g, E, w, -1), "GrLessInt")):
            for _ in range(5):
                fn()
            ts = []
            for _ in range(60):
                t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
            eng.profile(True); eng.profile_reset(); fn()
            kern = sum(eng.profile_read(k)[0] for k in bench.SCF_FAMILIES); eng.profile(False)
            print(f"n={N} {name:9s} m={m:4d}: median {np.median(ts)*1e6:7.1f} us  min {np.min(ts)*1e6:7.1f} us   kernel {kern*1e3:7.1f} us", flush=True)
    segs = [(rng.uniform(-2, 2, k) + 0.1j, np.ones(k)) for k in (2, 4, 12, 36)]
    for _ in range(5):
        GrIntSegments(F, S, g, segs)
    ts = []
    for _ in range(60):
        t = time.perf_counter(); GrIntSegments(F, S, g, segs); ts.append(time.perf_counter() - t)
    print(f"n={N} GrIntSegments 2+4+12+36: median {np.median(ts)*1e6:7.1f} us  min {np.min(ts)*1e6:7.1f} us", flush=True)
    segs = [(rng.uniform(-2, 2, k) + 0.1j, np.ones(k)) for k in (2, 4, 12, 36, 108, 324) * 2]      # a joint arc + tail probe, all levels
    for _ in range(5):
        GrIntSegments(F, S, g, segs)
    ts = []
    for _ in range(60):
        t = time.perf_counter(); GrIntSegments(F, S, g, segs); ts.append(time.perf_counter() - t)
    eng.profile(True); eng.profile_reset(); GrIntSegments(F, S, g, segs)
    kern = sum(eng.profile_read(k)[0] for k in bench.SCF_FAMILIES); eng.profile(False)
    print(f"n={N} GrIntSegments 972 points in 12 segments: median {np.median(ts)*1e6:7.1f} us  min {np.min(ts)*1e6:7.1f} us   kernel {kern*1e3:7.1f} us", flush=True)
