#!/usr/bin/env python3
"""
Time the BASELINE.json configurations C1..C5 on ONE GPU through the
drop-in Python API, with a bounded oracle (CPU) sample beside each.  Not the driver's bench
(bench.py is); this is the per-config table quoted in DESIGN.md.

    python scripts/bench_configs.py [c1 c2 c3 c4 c5] > gpurun_out/configs.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle                                             # noqa: E402
from helpers import chain_lead, random_system             # noqa: E402
from gaunegf_amd.engine import get_engine                 # noqa: E402
from gaunegf_amd.integrate import GrInt, GrLessInt        # noqa: E402
from gaunegf_amd.matTools import formSigma                # noqa: E402
from gaunegf_amd.surfGTester import surfGTest             # noqa: E402
from gaunegf_amd.surfG1D import surfG                     # noqa: E402
from gaunegf_amd.transport import SigmaCalculator, calculate_transmission   # noqa: E402
from gaunegf_amd import density as D                      # noqa: E402


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def cpu_rate(fn_sample, n_sample):
    t0 = time.perf_counter(); r = fn_sample(); dt = time.perf_counter() - t0
    return n_sample / dt, r


def c1():
    N, nc, M = 60, 15, 100
    F, S = random_system(N, 1)
    inds = [list(range(nc)), list(range(N - nc, N))]
    g = surfGTest(F, S, inds, -0.1j); ref = oracle.ConstSigma(F, S, inds, -0.1j)
    E, w = oracle.bias_window_grid(-0.25, 0.25, M, 0.0)
    out = {"config": "C1 ethane-sized surrogate: N=60, Gamma=0.2 eV constant contacts, 100 real-axis points"}
    for name, call, ocall in (("GrInt", lambda: GrInt(F, S, g, E, w), lambda: oracle.GrInt(F, S, ref, E, w)),
                              ("GrLessInt_ind-1", lambda: GrLessInt(F, S, g, E, w, -1), lambda: oracle.GrLessInt(F, S, ref, E, w, -1))):
        t, r = timed(call)
        rate, ro = cpu_rate(ocall, M)
        out[name] = {"gpu_s": t, "gpu_pts_per_s": M / t, "cpu_pts_per_s": rate, "rel_fro_vs_oracle": rel(r, ro)}
    return out


def c2():
    N, nc, M = 200, 20, 1000
    F, S = random_system(N, 2)
    inds = [list(range(nc)), list(range(N - nc, N))]
    g = surfGTest(F, S, inds, -0.1j); ref = oracle.ConstSigma(F, S, inds, -0.1j)
    E, w = oracle.real_axis_grid(-3.0, 3.0, M, 0.0); w = np.ones_like(E) * (6.0 / M)
    sub = slice(0, M, 20)
    out = {"config": "C2: N=200, constant Sigma, 1000 Legendre points on [-3,3] eV (host-pointer API, includes PCIe)"}
    t, r = timed(lambda: GrInt(F, S, g, E, w))
    rate, ro = cpu_rate(lambda: oracle.GrInt(F, S, ref, E[sub], w[sub]), len(E[sub]))
    out["GrInt"] = {"gpu_s": t, "gpu_pts_per_s": M / t, "cpu_pts_per_s": rate,
                    "rel_fro_vs_oracle_sample": rel(GrInt(F, S, g, E[sub], w[sub]), ro)}
    t, r = timed(lambda: GrLessInt(F, S, g, E, w, -1))
    rate, ro = cpu_rate(lambda: oracle.GrLessInt(F, S, ref, E[sub], w[sub], -1), len(E[sub]))
    out["GrLessInt_ind-1"] = {"gpu_s": t, "gpu_pts_per_s": M / t, "cpu_pts_per_s": rate,
                              "rel_fro_vs_oracle_sample": rel(GrLessInt(F, S, g, E[sub], w[sub], -1), ro)}
    sc = SigmaCalculator(g.sig[0], g.sig[1])
    t, T = timed(lambda: calculate_transmission(F, S, sc, E))
    st = sc.get_sigma_total(0); g1 = sc.get_gamma(0, 0); g2 = sc.get_gamma(0, -1)
    rate, Tr = cpu_rate(lambda: np.array([oracle.transmission_restricted(e, F, S, st, g1, g2) for e in E[sub]]), len(E[sub]))
    out["transmission"] = {"gpu_s": t, "gpu_pts_per_s": M / t, "cpu_pts_per_s": rate,
                           "max_rel_err_sample": float(np.max(np.abs(T[sub] - Tr) / np.maximum(1, np.abs(Tr))))}
    return out


def _c3_system(N=500, nc=50, eta=1e-4):
    F, S = random_system(N, 3)
    left = list(range(nc)); right = list(range(N - nc, N))
    aL = chain_lead(nc, 31); aR = chain_lead(nc, 32)
    kw = dict(taus=[aL[2].copy(), aR[2].copy()], staus=[aL[3].copy(), aR[3].copy()], alphas=[aL[0], aR[0]],
              aOverlaps=[aL[1], aR[1]], betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=eta)
    g = surfG(F, S, [left, right], **kw)
    ref = oracle.Chain1DSigma(F, S, [left, right], kw["taus"], kw["staus"], kw["alphas"], kw["aOverlaps"],
                              kw["betas"], kw["bOverlaps"], eta=eta)
    return F, S, g, ref


def c3():
    N, M = 500, 2000
    F, S, g, ref = _c3_system()
    E, w = oracle.real_axis_grid(-2.0, 2.0, M, 0.0); w = np.ones_like(E) * (4.0 / M)
    out = {"config": "C3: N=500, 1-D chain decimation (n_c=50, eta=1e-4), 2000 Legendre points on [-2,2] eV"}
    t, (sig, iters, conv) = timed(lambda: g.sigma_batch(E[:256]), reps=1)
    out["sigma_eval_256pts"] = {"gpu_s": t, "pts_per_s": 256 / t, "iters_mean": float(iters.mean()),
                                "iters_max": int(iters.max()), "converged_frac": float(conv.mean())}
    sub = np.arange(0, M, 200)
    rate, so = cpu_rate(lambda: np.stack([ref.sigmaTot(e) for e in E[sub]]), len(sub))
    out["sigma_eval_256pts"]["cpu_pts_per_s"] = rate
    out["sigma_eval_256pts"]["rel_fro_vs_oracle_sample"] = rel(g.sigma_batch(E[sub])[0], so)
    t, r = timed(lambda: GrInt(F, S, g, E, w), reps=1)
    rate, ro = cpu_rate(lambda: oracle.GrInt(F, S, ref, E[sub], w[sub]), len(sub))
    out["GrInt"] = {"gpu_s": t, "gpu_pts_per_s": M / t, "cpu_pts_per_s": rate,
                    "rel_fro_vs_oracle_sample": rel(GrInt(F, S, g, E[sub], w[sub]), ro)}
    return out


def _bethe_contacts(N):
    """Geometry of C4's contacts: 2 contacts x 3 Au atoms x 9 orbitals at the two ends of
    an N-orbital device; the remaining orbitals belong to one device 'atom'."""
    coords = np.array([[0, 0, 0.0], [2.88, 0, 0], [1.44, 2.494, 0],
                       [0, 0, 20.0], [2.88, 0, 20.0], [1.44, 2.494, 20.0],
                       [1.44, 0.8, 10.0]])
    orbMap = np.concatenate([np.full(9, a + 1) for a in range(6)] + [np.full(N - 54, 7)])
    typ_one = np.array([0, 1001, 1002, 1003, 2001, 2002, 2003, 2004, 2005])
    orbTyp = np.concatenate([typ_one] * 6 + [np.zeros(N - 54, dtype=int)])
    return coords, orbMap, orbTyp


def c4():
    from gaunegf_amd.surfGBethe import surfGB
    N = 800
    F, S = random_system(N, 4)
    coords, orbMap, orbTyp = _bethe_contacts(N)
    lat = os.path.join(ROOT, "gaunegf_amd", "data", "Au")
    t0 = time.perf_counter()
    g = surfGB.from_arrays(F, S, [[1, 2, 3], [4, 5, 6]], orbMap, orbTyp, coords, latFile=lat, eta=1e-6, fermi=0.0)
    t_setup = time.perf_counter() - t0
    out = {"config": "C4: N=800, Bethe-lattice Sigma (Au.bethe, 2 contacts x 3 atoms x 9 orbitals, eta=1e-6), "
                     "contour ANT N=486 (T=0) + real axis N2=256",
           "host_setup_s": t_setup, "orthogonalised_Sigma": bool(g.Sdict['sss'] == 0)}
    t, P = timed(lambda: D.densityComplexN(F, S, g, -8.0, 0.0, 486, 0.0, showText=False), reps=1)
    out["densityComplexN_486"] = {"gpu_s": t, "gpu_pts_per_s": 486 / t, "gpu_tflops_inverse_only": 8.0 * N ** 3 * 486 / t / 1e12}
    t, P2 = timed(lambda: D.densityRealN(F, S, g, -1e6, -8.0, 256, 0.0, showText=False), reps=1)
    out["densityRealN_256"] = {"gpu_s": t, "gpu_pts_per_s": 256 / t}
    # oracle sample: 3 contour points with the oracle's own Bethe fixed point at the same trip count
    E, w = oracle.contour_grid(-8.0, 0.0, 486, 0.0)
    sub = np.array([0, 243, 485])
    Xi = g.Xi if g.Sdict['sss'] == 0 else None
    g.force_iters = 30

    class Ref:
        def sigma(self, E_, i, conv=None):
            at = g.gList[i]
            return oracle.bethe_contact_sigma(E_, N, g.indsLists[i], g.nIndLists[i], at.H, at.Slist, at.Vlist,
                                              1e-6, Xi=Xi, force_iters=30)
        def sigmaTot(self, E_, conv=None): return self.sigma(E_, 0) + self.sigma(E_, 1)
    rate, ro = cpu_rate(lambda: oracle.GrInt(F, S, Ref(), E[sub], w[sub]), len(sub))
    out["densityComplexN_486"]["cpu_pts_per_s"] = rate
    out["densityComplexN_486"]["rel_fro_vs_oracle_sample_fixed_trip"] = rel(GrInt(F, S, g, E[sub], w[sub]), ro)
    g.force_iters = -1
    return out


def c4b():
    """Bethe-lattice surface self-energy (the Sigma of C4): one Au contact atom, 486 contour + 256
    real-axis energies, free-running fixed points (bulk 12 directions, then surface 6 directions)."""
    import os
    from gaunegf_amd.surfGBethe import read_bethe_params, construct_sk_matrix, gen_neighbors, surfGBAt
    here = os.path.join(ROOT, "gaunegf_amd", "data", "Au")
    ne, Ed, Vd, Sd, H0 = read_bethe_params(here)
    dirs = gen_neighbors(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.2, 0.0]))
    Sl = [construct_sk_matrix(Sd, d) for d in dirs]; Vl = [construct_sk_matrix(Vd, d) for d in dirs]
    at = surfGBAt(H0, Sl, Vl, 1e-6)
    Ec, _ = oracle.contour_grid(-30.0, 0.0, 486, 0.0)
    Er = np.linspace(-1.0, 1.0, 256)
    out = {"config": "C4 Sigma: Bethe lattice, Au parameters, eta=1e-6, 486 complex-contour + 256 real-axis energies"}
    for name, E in (("contour_486", np.asarray(Ec)), ("real_axis_256", Er)):
        at.sigma(E[:4])
        t0 = time.perf_counter(); sig = at.sigma(E); t = time.perf_counter() - t0
        its = np.asarray(at.last_iters)
        bulk, surf = its & 0xFFFF, its >> 16           # sweeps of the bulk and of the surface fixed point
        out[name] = {"gpu_s": t, "pts_per_s": len(E) / t, "bulk_sweeps_mean": float(np.mean(bulk)),
                     "surface_sweeps_mean": float(np.mean(surf)), "sweeps_max": int(max(np.max(bulk), np.max(surf)))}
    e = complex(Er[100])
    t0 = time.perf_counter(); ref = oracle.bethe_sigma_surface(e, H0, Sl, Vl, 1e-6); tc = time.perf_counter() - t0
    out["real_axis_256"]["cpu_pts_per_s"] = 1.0 / tc
    out["real_axis_256"]["rel_fro_vs_oracle_sample"] = rel(at.sigma(np.array([e]))[0], ref[0])
    return out


def c5():
    """C5 through the drop-in API: the 2N x 2N spin-'u' system is block diagonal (scf.py:177-180), so
    GrLessInt and the spin-block transmission run as two N = 1000 solves (integrate._spin_split,
    transport._transmission_batch); the full 2N x 2N path is timed beside it (SPIN_BLOCK_SPLIT off)."""
    import gaunegf_amd.integrate as I
    import gaunegf_amd.transport as T
    N = 1000
    Fa, Sa = random_system(N, 5); Fb, _ = random_system(N, 6)
    F = np.block([[Fa, np.zeros((N, N))], [np.zeros((N, N)), Fb]]); S = np.kron(np.eye(2), Sa)
    nc = 30
    left = list(range(nc)); right = list(range(N - nc, N))
    s1 = formSigma(left, -0.1j, N, Sa); s2 = formSigma(right, -0.1j, N, Sa)
    inds2 = [left + [N + i for i in left], right + [N + i for i in right]]
    g = surfGTest(F, S, inds2, -0.1j)                       # constant provider on the 2N space: kron(I2, sigma)
    assert np.allclose(g.sig[0], np.kron(np.eye(2), s1))
    ref = oracle.ConstSigma(F, S, inds2, -0.1j)
    sc = SigmaCalculator(s1, s2)
    M = 256
    E, w = oracle.bias_window_grid(-0.25, 0.25, M, 300.0)
    Et = np.linspace(-0.25, 0.25, M)
    out = {"config": f"C5 (1 GPU share): 2 x 1000 spin-block F/S (2000 x 2000), qV=0.5 V window, T=300 K, "
                     f"{M} of the 512 Legendre points; drop-in API (host pointers, incl. PCIe)"}
    for split in (True, False):
        I.SPIN_BLOCK_SPLIT = T.SPIN_BLOCK_SPLIT = split
        try:
            tag = "two_N_solves" if split else "full_2N"
            t, r = timed(lambda: GrLessInt(F, S, g, E, w, -1), reps=1)
            t2, (Tt, Ts) = timed(lambda: calculate_transmission(F, S, sc, Et, spin='u'), reps=1)
            out[tag] = {"GrLessInt_ind-1_s": t, "GrLessInt_pts_per_s": M / t,
                        "GrLessInt_tflops_of_the_2N_problem": 24.0 * (2 * N) ** 3 * M / t / 1e12,
                        "transmission_spin_u_s": t2, "transmission_pts_per_s": M / t2}
            if split:
                r_split, T_split = r, Tt
            else:
                out["split_vs_full_rel_fro"] = rel(r_split, r)
                out["split_vs_full_T_max_rel"] = float(np.max(np.abs(T_split - Tt) / np.maximum(1e-300, np.abs(Tt))))
        finally:
            I.SPIN_BLOCK_SPLIT = T.SPIN_BLOCK_SPLIT = True
    sub = [0, M // 2]
    rate, ro = cpu_rate(lambda: oracle.GrLessInt(F, S, ref, E[sub], w[sub], -1), len(sub))
    out["cpu_pts_per_s"] = rate
    out["rel_fro_vs_oracle_sample"] = rel(GrLessInt(F, S, g, E[sub], w[sub], -1), ro)
    return out


if __name__ == "__main__":
    want = sys.argv[1:] or ["c1", "c2", "c3", "c4", "c4b", "c5"]
    res = {}
    for name in want:
        t0 = time.perf_counter()
        try:
            res[name] = globals()[name]()
        except Exception as e:                           # keep going, report the failure
            res[name] = {"error": repr(e)}
        res[name]["wall_s"] = time.perf_counter() - t0
        print(f"# {name} done in {res[name]['wall_s']:.1f} s", file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))
