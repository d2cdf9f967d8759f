#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by EXECUTING the reference's own
numpy-only functions in this container (the reference checkout lives at
/root/reference and never travels; only the vectors below are committed).

Why AST extraction: every hot-path module of the reference does ``import jax`` (or
``import gauopen``) at module level and neither package is installed here, so the
modules cannot be imported (ordinary ModuleNotFoundError).
Many functions inside them are nevertheless pure numpy/scipy.  This script parses
the reference source with ``ast``, pulls out those functions/classes BY NAME and
executes them -- unmodified -- in a namespace holding the real numpy/scipy and the
reference's own config constants.  No arithmetic stand-in for jax is involved: a
function that touches jax is simply not extracted.

Two kinds of vectors are produced:
  (1) values computed entirely by reference code (ANT nodes, fermi, formSigma,
      SigmaCalculator, the reference tests' numpy loops serial_gr_integration /
      serial_gless_integration / manual_iteration on the reference tests' seeded
      generators);
  (2) "boundary captures": the reference's grid builders (density.densityRealN,
      densityGridN, densityComplexN, densityComplex, densityGrid,
      transport.calculate_current) are run with a SPY bound to the name of the
      engine entry point they call (GrInt / GrLessInt / calculate_transmission).
      The spy records the (Elist, weights[, ind]) the reference hands to the engine
      -- i.e. exactly the arrays that cross the drop-in boundary -- and returns a
      cheap analytic matrix so the adaptive drivers walk their level sequence.

Run:  python tests/golden/make_golden.py      (needs /root/reference)
"""
import ast
import importlib.util
import io
import os
import sys
import contextlib

import numpy as np
import scipy
from scipy.special import roots_legendre, roots_chebyu
from scipy.integrate import trapezoid

REF = os.environ.get("GAUNEGF_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def extract(path, names, ns):
    """exec the top-level defs/classes/assignments called ``names`` from ``path`` into ns."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    want = set(names)
    found = set()
    body = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in want:
            # drop decorators (``@jit`` would need jax); none of the extracted
            # functions is decorated, assert that instead of silently stripping
            assert not getattr(node, "decorator_list", []), (path, node.name)
            body.append(node); found.add(node.name)
        elif isinstance(node, ast.Assign):
            tg = [t.id for t in node.targets if isinstance(t, ast.Name)]
            if tg and all(t in want for t in tg):
                body.append(node); found.update(tg)
    missing = want - found
    assert not missing, f"{path}: not found {missing}"
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, os.path.join(REF, path), "exec"), ns)


def load_config():
    spec = importlib.util.spec_from_file_location("ref_config", os.path.join(REF, "gauNEGF/config.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    cfg = load_config()
    out = {}

    # ------------------------------------------------------------------ config
    out["config_names"] = np.array(sorted(k for k in vars(cfg) if k.isupper()))
    out["config_values"] = np.array([str(getattr(cfg, k)) for k in out["config_names"]])

    # ------------------------------------------------------- density.py (numpy)
    dns = {"np": np, "roots_legendre": roots_legendre, "roots_chebyu": roots_chebyu}
    for k in ("TEMPERATURE", "ADAPTIVE_INTEGRATION_TOL", "FERMI_CALCULATION_TOL",
              "FERMI_SEARCH_CYCLES", "N_KT", "ENERGY_MIN", "MAX_CYCLES", "MAX_GRID_POINTS"):
        dns[k] = getattr(cfg, k)
    extract("gauNEGF/density.py",
            ["har_to_eV", "kB", "fermi", "getANTPoints", "integratePointsAdaptiveANT",
             "densityRealN", "densityReal", "densityGridN", "densityGrid",
             "densityComplexN", "densityComplex"], dns)

    for N in (2, 6, 18, 54, 100):
        x, w = dns["getANTPoints"](N)
        out[f"ant_x_{N}"] = x; out[f"ant_w_{N}"] = w

    # tests/testANT.py main loop, value by value (its getANTPoints is a copy)
    tns = {"np": np}
    extract("tests/testANT.py", ["getANTPoints"], tns)
    func = lambda x: np.exp(-x ** 2)
    direct = []
    for N in (2, 6, 18, 54, 162, 486):
        x, w = tns["getANTPoints"](N)
        direct.append(float(np.dot(w, func(x))))
    out["testant_direct"] = np.array(direct)

    Es = np.array([-2.0, -0.5, 0.0, 0.25, 0.5, 3.0])
    Ez = np.array([-1 + 0.5j, 0.3 + 0.0j, 0.3 + 1e-3j, 0.3 - 1e-3j, 0.31 + 5j, 0.29 - 5j])
    out["fermi_E_real"] = Es; out["fermi_E_cplx"] = Ez
    out["fermi_T0_real"] = np.asarray(dns["fermi"](Es, 0.3, 0))
    out["fermi_T0_cplx"] = np.asarray(dns["fermi"](Ez, 0.3, 0))
    out["fermi_T300_real"] = np.asarray(dns["fermi"](Es, 0.3, 300.0))
    out["fermi_T300_cplx"] = np.asarray(dns["fermi"](Ez, 0.3, 300.0))

    # ---- boundary captures: what the reference passes to GrInt / GrLessInt ----
    Nf = 3
    F = np.diag([-1.0, 0.2, 0.9]); S = np.eye(Nf)
    poles = np.array([-0.7 - 0.05j, 0.1 - 0.2j, 0.6 - 0.01j])
    calls = []

    def spy_GrInt(F_, S_, g_, Elist, weights):
        Elist = np.asarray(Elist); weights = np.asarray(weights)
        calls.append(("GrInt", Elist.copy(), weights.copy(), None))
        acc = np.zeros((Nf, Nf), dtype=complex)
        for E, w in zip(Elist, weights):
            acc += w * np.diag(1.0 / (E - poles))
        return acc

    def spy_GrLessInt(F_, S_, g_, Elist, weights, ind=None):
        Elist = np.asarray(Elist); weights = np.asarray(weights)
        calls.append(("GrLessInt", Elist.copy(), weights.copy(), ind))
        acc = np.zeros((Nf, Nf), dtype=complex)
        for E, w in zip(Elist, weights):
            acc += w * np.diag(np.abs(1.0 / (E - poles)) ** 2)
        return acc

    dns["GrInt"] = spy_GrInt
    dns["GrLessInt"] = spy_GrLessInt

    def capture(tag, fn, *a, **k):
        calls.clear()
        res = quiet(fn, *a, **k)
        out[f"{tag}_ncalls"] = np.array(len(calls))
        for i, (name, E, w, ind) in enumerate(calls):
            out[f"{tag}_c{i}_name"] = np.array(name)
            out[f"{tag}_c{i}_E"] = E
            out[f"{tag}_c{i}_w"] = w
            out[f"{tag}_c{i}_ind"] = np.array(-99 if ind is None else ind)
        out[f"{tag}_result"] = np.asarray(res)

    capture("realN_T0", dns["densityRealN"], F, S, None, -3.0, 0.3, 24, 0.0, showText=False)
    capture("realN_T300", dns["densityRealN"], F, S, None, -3.0, 0.3, 17, 300.0, showText=False)
    capture("gridN_T0_fwd", dns["densityGridN"], F, S, None, -0.25, 0.25, -1, 16, 0.0, showText=False)
    capture("gridN_T300_rev", dns["densityGridN"], F, S, None, 0.25, -0.25, 0, 20, 300.0, showText=False)
    capture("gridN_T300_none", dns["densityGridN"], F, S, None, -0.1, 0.4, None, 9, 300.0, showText=False)
    for meth in ("ant", "legendre", "chebyshev", "midpoint"):
        capture(f"cplxN_T0_{meth}", dns["densityComplexN"], F, S, None, -5.0, 0.3, 18, 0.0,
                showText=False, method=meth)
        capture(f"cplxN_T300_{meth}", dns["densityComplexN"], F, S, None, -5.0, 0.3, 32, 300.0,
                showText=False, method=meth)
    capture("cplx_adapt_T0", dns["densityComplex"], F, S, None, -5.0, 0.3, 1e-6, 0.0)
    capture("cplx_adapt_T300", dns["densityComplex"], F, S, None, -5.0, 0.3, 1e-5, 300.0)
    capture("grid_adapt_T300", dns["densityGrid"], F, S, None, -0.25, 0.25, -1, 1e-6, 300.0)
    capture("real_adapt_T0", dns["densityReal"], F, S, None, -3.0, 0.3, 1e-3, 0.0, 200)

    # ------------------------------------------------------ matTools.formSigma
    mns = {"np": np}
    extract("gauNEGF/matTools.py", ["formSigma"], mns)
    rng = np.random.default_rng(7)
    Ssm = np.eye(6) + 0.05 * (lambda a: a + a.T)(rng.standard_normal((6, 6)))
    Vm = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
    out["formsigma_S"] = Ssm; out["formsigma_Vm"] = Vm
    out["formsigma_scalar"] = mns["formSigma"]([0, 1], -0.1j, 6, Ssm)
    out["formsigma_matrix"] = mns["formSigma"]([4, 5], Vm, 6, Ssm)
    out["formsigma_noS"] = mns["formSigma"]([2], -0.05j, 6)

    # ------------------------------------------------- transport.SigmaCalculator
    trn = {"np": np, "os": os, "trapezoid": trapezoid,
           "ENERGY_STEP": cfg.ENERGY_STEP, "N_KT": cfg.N_KT, "TEMPERATURE": cfg.TEMPERATURE}
    extract("gauNEGF/transport.py",
            ["har_to_eV", "eoverh", "kB", "V_to_au", "SigmaCalculator", "calculate_current"], trn)
    s1 = out["formsigma_scalar"]; s2 = out["formsigma_matrix"]
    sc = trn["SigmaCalculator"](s1, s2)
    out["sc_tot"] = np.asarray(sc.get_sigma_total(0.1))
    out["sc_tot_u"] = np.asarray(sc.get_sigma_total(0.1, 'u', 12))
    out["sc_tot_g"] = np.asarray(sc.get_sigma_total(0.1, 'g', 12))
    out["sc_gam0"] = np.asarray(sc.get_gamma(0.1, 0))
    out["sc_gam1_u"] = np.asarray(sc.get_gamma(0.1, -1, 'u', 12))
    v1 = np.array([-0.1j, -0.1j, 0, 0, 0, 0]); v2 = np.array([0, 0, 0, 0, -0.2j, -0.2j])
    scv = trn["SigmaCalculator"](v1, v2)
    out["scv_tot"] = np.asarray(scv.get_sigma_total(0.0))
    out["scv_gam1"] = np.asarray(scv.get_gamma(0.0, 1))

    # calculate_current: capture the np.arange grid and the quadrature
    tcalls = []

    def spy_calc_T(F_, S_, sc_, energies, spin=None, **kw):
        energies = np.asarray(energies)
        tcalls.append(energies.copy())
        T = 1.0 / (1.0 + (energies - 0.05) ** 2)
        if spin in ('u', 'ro', 'g'):
            return T, np.stack([0.4 * T, 0.1 * T, 0.1 * T, 0.4 * T], axis=1)
        return T
    trn["calculate_transmission"] = spy_calc_T
    for tag, args in {
        "cur_T0_pos": dict(fermi=0.1, qV=0.05, T=0.0, spin='r', dE=0.001),
        "cur_T0_neg": dict(fermi=0.1, qV=-0.05, T=0.0, spin='r', dE=0.001),
        "cur_T300_pos": dict(fermi=-0.2, qV=0.1, T=300.0, spin='r', dE=0.002),
        "cur_T300_neg_u": dict(fermi=-0.2, qV=-0.1, T=300.0, spin='u', dE=0.002),
    }.items():
        tcalls.clear()
        res = trn["calculate_current"](F, S, None, **args)
        out[f"{tag}_grid"] = tcalls[0]
        if isinstance(res, tuple):
            out[f"{tag}_value"] = np.array(res[0]); out[f"{tag}_spin"] = np.array(res[1])
        else:
            out[f"{tag}_value"] = np.array(res)
    out["cur_zero_r"] = np.array(trn["calculate_current"](F, S, None, 0.0, 0.0))

    np.savez_compressed(os.path.join(OUT, "ref_bookkeeping.npz"), **out)
    print("ref_bookkeeping.npz:", len(out), "arrays")

    # ------------------------------------- reference tests' numpy restatements
    out = {}
    cns = {"np": np}
    extract("tests/test_computation_consistency.py",
            ["MockSurfaceGreen", "create_test_matrices", "serial_gr_integration",
             "serial_gless_integration"], cns)
    for size in (12, 40):
        Fm, Sm = cns["create_test_matrices"](size)
        g = cns["MockSurfaceGreen"](size)
        # grid exactly as tests/test_computation_consistency.py:250-252
        num_energies = 12
        Elist = np.linspace(-1.0, 1.0, num_energies) + 1j * 0.01
        weights = np.ones(num_energies, dtype=complex) * (Elist[1] - Elist[0])
        out[f"cc{size}_F"] = Fm; out[f"cc{size}_S"] = Sm
        out[f"cc{size}_sigma_base"] = g._sigma_base
        out[f"cc{size}_sigma_c0"] = g._sigma_contacts[0]
        out[f"cc{size}_sigma_c1"] = g._sigma_contacts[1]
        out[f"cc{size}_E"] = Elist; out[f"cc{size}_w"] = weights
        out[f"cc{size}_gr"] = cns["serial_gr_integration"](Fm, Sm, g, Elist, weights)
        out[f"cc{size}_gless_none"] = cns["serial_gless_integration"](Fm, Sm, g, Elist, weights, None)
        out[f"cc{size}_gless_0"] = cns["serial_gless_integration"](Fm, Sm, g, Elist, weights, 0)
        out[f"cc{size}_gless_1"] = cns["serial_gless_integration"](Fm, Sm, g, Elist, weights, 1)

    sns = {"np": np}
    extract("tests/test_surface_green_jit.py", ["create_test_contact", "manual_iteration"], sns)
    energies = [0.0, 1.0, -1.0, 0.5j]     # tests/test_surface_green_jit.py:76
    out["sg_energies"] = np.array(energies, dtype=complex)
    for size in (5, 10, 20):
        alpha, Salpha, beta, Sbeta = sns["create_test_contact"](size)
        out[f"sg{size}_alpha"] = alpha; out[f"sg{size}_Salpha"] = Salpha
        out[f"sg{size}_beta"] = beta; out[f"sg{size}_Sbeta"] = Sbeta
        for ie, E in enumerate(energies):
            eta = 1e-4
            A = (E + 1j * eta) * Salpha - alpha
            B = (E + 1j * eta) * Sbeta - beta
            g0 = np.zeros_like(A, dtype=complex)
            gm, conv_flag, iters = sns["manual_iteration"](A, B, g0, 1e-8, 0.1, 500)
            out[f"sg{size}_e{ie}_g"] = gm
            out[f"sg{size}_e{ie}_conv"] = np.array(conv_flag)
            out[f"sg{size}_e{ie}_iters"] = np.array(iters)
            # the PRODUCTION start of the loop, g0 = inv(A) (surfG1D.py:287: solve(A, I)), handed to the same
            # reference function: pins the start the product and the oracle use when no g_init is given
            g0p = np.linalg.solve(A, np.eye(A.shape[0]))
            gm, conv_flag, iters = sns["manual_iteration"](A, B, g0p, 1e-8, 0.1, 500)
            out[f"sg{size}_e{ie}_g_invstart"] = gm
            out[f"sg{size}_e{ie}_conv_invstart"] = np.array(conv_flag)
            out[f"sg{size}_e{ie}_iters_invstart"] = np.array(iters)

    bns = {"np": np, "os": os}
    extract("tests/benchmark_sigma_parallelization.py",
            ["SURFACE_GREEN_CONVERGENCE", "SURFACE_RELAXATION_FACTOR", "ETA",
             "configure_blas_threads", "manual_surface_green_iteration",
             "compute_sigma_for_energy"], bns)
    alpha, Salpha, beta, Sbeta = sns["create_test_contact"](8)
    env_before = dict(os.environ)
    for ie, E in enumerate([0.3, -0.8]):
        sig, conv_flag, iters = bns["compute_sigma_for_energy"](E, alpha, Salpha, beta, Sbeta, eta=1e-3)
        out[f"bs8_e{ie}_sigma"] = sig
        out[f"bs8_e{ie}_conv"] = np.array(conv_flag)
        out[f"bs8_e{ie}_iters"] = np.array(iters)
    os.environ.clear(); os.environ.update(env_before)
    out["bs8_energies"] = np.array([0.3, -0.8])
    out["bs8_alpha"] = alpha; out["bs8_Salpha"] = Salpha
    out["bs8_beta"] = beta; out["bs8_Sbeta"] = Sbeta

    wns = {"np": np}
    extract("tests/test_transport_checkpointing.py",
            ["create_nanowire_hamiltonian", "create_nanowire_with_contacts",
             "create_energy_independent_sigma_simple"], wns)
    Fw, Sw = wns["create_nanowire_hamiltonian"](20)
    out["wire20_F"] = Fw; out["wire20_S"] = Sw
    sgl, sgr = wns["create_energy_independent_sigma_simple"](20, list(range(10)), list(range(10, 20)), 0.1)
    out["wire20_sig1"] = sgl; out["wire20_sig2"] = sgr

    ons = {"np": np}
    extract("tests/jax_optimization_suite.py", ["create_realistic_transport_system"], ons)
    Fo, So, st, G1, G2 = ons["create_realistic_transport_system"](24)
    out["rts24_F"] = Fo; out["rts24_S"] = So; out["rts24_sigma"] = st
    out["rts24_G1"] = G1; out["rts24_G2"] = G2
    # BUILDER RESTATEMENT (not reference output): the check of tests/jax_optimization_suite.py:165-194
    # inlines its numpy twin inside a test function that also calls the jax kernels, so it cannot be
    # extracted on its own; the same two expressions are written out here (np.linalg.inv, trace of the
    # triple product).  The inputs above (rts24_F ... rts24_G2) ARE reference output.
    es = np.linspace(-2, 2, 7)
    Ts, Ds = [], []
    for E in es:
        mat = E * So - Fo - st
        Gr = np.linalg.inv(mat)
        Ts.append(np.real(np.trace(G1 @ Gr @ G2 @ np.conj(Gr).T)))
        Ds.append(-np.imag(np.trace(Gr)) / np.pi)
    out["rts24_energies"] = es
    out["rts24_T"] = np.array(Ts); out["rts24_dos"] = np.array(Ds)

    np.savez_compressed(os.path.join(OUT, "ref_numpy_restatements.npz"), **out)
    print("ref_numpy_restatements.npz:", len(out), "arrays")

    # ------------------------------------------- Bethe lattice: the reference's numpy twin (surfG3D.py)
    # gauNEGF/surfG3D.py is a numpy-only copy of surfGBethe.py's parser, neighbour generator and
    # Slater-Koster construction (:137-385), of the SURFACE fixed point (surfGAt.sigma, :907-979) and of
    # the 13-site cluster assembly (surfGAt.sigmaTot, :998-1031).  Its bulk loop (surfGAt.sigmaK) is a
    # different, Jacobi-type iteration than surfGBethe.py:958-1030 and is NOT used: the bulk self-energies
    # the surface loop starts from are INJECTED (a fixed, seeded set), so that exactly the code of the
    # surface loop / the cluster assembly is what produces the vectors.
    out = {}
    from numpy import linalg as LA
    g3 = {"np": np, "LA": LA, "ETA": cfg.ETA, "TEMPERATURE": cfg.TEMPERATURE, "ENERGY_MIN": cfg.ENERGY_MIN}
    extract("gauNEGF/surfG3D.py", ["kB", "dim", "har_to_eV", "Eminf", "surfG3", "surfGAt"], g3)
    for name in ("Au", "Au2"):
        dev = object.__new__(g3["surfG3"])                 # methods only: the constructor needs Gaussian
        dev.readBetheParams(os.path.join(REF, name))
        out[f"{name}_ne"] = np.array(dev.ne)
        out[f"{name}_H0"] = dev.H0
        for dname, d in (("E", dev.Edict), ("V", dev.Vdict), ("S", dev.Sdict)):
            keys = sorted(d)
            out[f"{name}_{dname}_keys"] = np.array(keys)
            out[f"{name}_{dname}_vals"] = np.array([d[k] for k in keys])
        geoms = [(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.2, 0.0])),
                 (np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0), np.array([0.3, -1.0, 0.4]))]
        for gi, (normal, first) in enumerate(geoms):
            dirs = dev.genNeighbors(normal, first.copy())
            out[f"{name}_g{gi}_normal"] = normal; out[f"{name}_g{gi}_first"] = first
            out[f"{name}_g{gi}_dirs"] = np.array(dirs)
            out[f"{name}_g{gi}_Slist"] = np.array([dev.constructMat(dev.Sdict, d) for d in dirs])
            out[f"{name}_g{gi}_Vlist"] = np.array([dev.constructMat(dev.Vdict, d) for d in dirs])
        Sl = list(out[f"{name}_g0_Slist"]); Vl = list(out[f"{name}_g0_Vlist"])
        eta = 1e-6
        at = g3["surfGAt"](dev.H0.copy(), [m.copy() for m in Sl], [m.copy() for m in Vl], eta)
        out[f"{name}_cluster_F"] = at.F; out[f"{name}_cluster_S"] = at.S
        rng = np.random.default_rng(11)
        energies = np.array([-5.0, 0.7, 2.5])
        out[f"{name}_energies"] = energies
        out[f"{name}_eta"] = np.array(eta)
        for ie, E in enumerate(energies):
            # a fixed, physically shaped bulk set: -i I plus a seeded symmetric perturbation
            pert = rng.standard_normal((12, 9, 9)) * 0.05
            sigK = np.array([-1j * np.eye(9) + (p_ + p_.T) * (0.3 - 0.2j) for p_ in pert])
            at.sigmaK = (lambda sk: (lambda E_, conv=1e-5, mix=0.5: sk.copy()))(sigK)
            out[f"{name}_e{ie}_sigK"] = sigK
            surf = quiet(at.sigma, float(E), None, 1e-5, 0.5)
            out[f"{name}_e{ie}_surface"] = np.array(surf)
            out[f"{name}_e{ie}_cluster"] = quiet(at.sigmaTot, float(E), 1e-5)
    # ---- contact assembly: surfG3.sigma / sigmaTot (surfG3D.py:417-433, 435-463), the numpy
    # twin of surfGBethe.py:479-575.  The object is created without its constructor (which needs Gaussian);
    # everything the two methods read is injected: N, the per-contact atoms' orbital index lists and attached
    # neighbour directions, the spin mode, and per contact an object whose sigma(E, i, conv) returns a given
    # [9,9,9] set of surface self-energies (a seeded set -- the assembly is what is pinned here, the surface
    # loop is pinned above).  Only lattices with Ssss != 0 (Au): for Ssss == 0 the twin calls a helper
    # ``times`` that no module of the reference defines (NameError), so the Xi branch cannot be executed.
    # Neighbour lists are subsets of 0..8 without repeats (the twin de-duplicates through set(), the jax
    # version subtracts every listed entry).
    class _GivenSurface:
        def __init__(self, sig9):
            self.sig9 = sig9
        def sigma(self, E, i, conv):
            return self.sig9.copy()
    devA = object.__new__(g3["surfG3"])
    devA.readBetheParams(os.path.join(REF, "Au"))
    Nasm = 40
    rng = np.random.default_rng(31)
    inds_lists = [[list(range(0, 9)), list(range(9, 18))], [list(range(22, 31)), list(range(31, 40))]]
    nind_lists = [[[0, 1, 2], [6, 7, 8, 3]], [[2, 5], []]]
    sig9 = [np.array([(lambda p_: -1j * np.eye(9) * (0.5 + 0.1 * k) + (p_ + p_.T) * (0.2 - 0.1j))(rng.standard_normal((9, 9)) * 0.1)
                      for k in range(9)]) for _ in range(2)]
    out["asm_N"] = np.array(Nasm)
    for c in range(2):
        out[f"asm_c{c}_sig9"] = sig9[c]
        for a_, (fi_, ni_) in enumerate(zip(inds_lists[c], nind_lists[c])):
            out[f"asm_c{c}_a{a_}_inds"] = np.array(fi_)
            out[f"asm_c{c}_a{a_}_nInds"] = np.array(ni_, dtype=int)
    devA.N = Nasm
    devA.gList = [_GivenSurface(sig9[0]), _GivenSurface(sig9[1])]
    devA.indsLists = inds_lists
    devA.nIndLists = nind_lists
    devA.Xi = None
    assert devA.Sdict['sss'] != 0
    for spin in ("r", "u", "g"):
        devA.spin = spin
        out[f"asm_{spin}_sigma0"] = devA.sigma(0.3, 0)
        out[f"asm_{spin}_sigma1"] = devA.sigma(0.3, 1)
        out[f"asm_{spin}_sigmaTot"] = devA.sigmaTot(0.3)

    np.savez_compressed(os.path.join(OUT, "ref_bethe.npz"), **out)
    print("ref_bethe.npz:", len(out), "arrays")

    # ------------------------- analytic density for constant self-energies (density.py:276-382, numpy only)
    out = {}
    ans = {"np": np, "FERMI_CALCULATION_TOL": cfg.FERMI_CALCULATION_TOL, "ENERGY_MIN": cfg.ENERGY_MIN}
    extract("gauNEGF/density.py", ["density", "bisectFermi"], ans)
    from scipy.linalg import fractional_matrix_power
    rng = np.random.default_rng(21)
    Na = 10
    A_ = rng.standard_normal((Na, Na)); Fa = (A_ + A_.T) * (1.0 / np.sqrt(2 * Na)) * 2
    B_ = rng.standard_normal((Na, Na)); Sa = np.eye(Na) + 0.1 * (B_ + B_.T) / np.sqrt(2 * Na)
    s1 = mns["formSigma"]([0, 1], -0.1j, Na, Sa); s2 = mns["formSigma"]([8, 9], -0.15j, Na, Sa)
    X = np.array(fractional_matrix_power(Sa, -0.5))
    Fbar = X @ (Fa + s1 + s2) @ X
    Gam = X @ (1j * (s1 - s1.conj().T) + 1j * (s2 - s2.conj().T)) @ X
    Dv, Vv = np.linalg.eig(Fbar)
    Vc = np.linalg.inv(Vv.conj().T)
    out["an_F"] = Fa; out["an_S"] = Sa; out["an_sig1"] = s1; out["an_sig2"] = s2
    out["an_D"] = Dv; out["an_V"] = Vv; out["an_Vc"] = Vc; out["an_Gam"] = Gam; out["an_X"] = X
    for i, (Emin_, mu_) in enumerate([(-1e6, 0.2), (-40.0, -0.5), (-6.0, 1.0)]):
        out[f"an{i}_limits"] = np.array([Emin_, mu_])
        out[f"an{i}_Pbar"] = ans["density"](Vv, Vc, Dv, Gam, Emin_, mu_)
    out["an_bisect_Nexp"] = np.array(4.3)
    out["an_bisect_fermi"] = np.array(quiet(ans["bisectFermi"], Vv, Vc, Dv, Gam, 4.3, 1e-6, -1e6))
    np.savez_compressed(os.path.join(OUT, "ref_analytic_density.npz"), **out)
    print("ref_analytic_density.npz:", len(out), "arrays")
    # ------------------ Fermi searches / integration-limit fitting (SURVEY f-1; density.py:821-1515, numpy bodies)
    # The reference's own function bodies, AST-extracted and executed unmodified; the callables they use
    # (densityComplexN / densityComplex / densityRealN / densityGridN, _compute_dos_at_energy, g.setF, g.sigmaTot)
    # are the analytic spies of tests/fermi_probe_harness.py, which record every probe.  ``inv`` / ``eigh`` are
    # bound to numpy's solve-with-identity / eigh: gauNEGF/utils.py:52-62 defines them as exactly those one-line
    # jnp calls (calcEmin is the only user).
    sys.path.insert(0, os.path.dirname(OUT))
    import fermi_probe_harness as H
    out = {}
    rec = []
    fns = {"np": np, "inv": lambda A: np.linalg.solve(A, np.eye(A.shape[0])), "eigh": np.linalg.eigh,
           "FERMI_DEBUG": False}
    for k in ("TEMPERATURE", "ADAPTIVE_INTEGRATION_TOL", "FERMI_CALCULATION_TOL", "FERMI_SEARCH_CYCLES",
              "ENERGY_MIN", "MAX_CYCLES", "MAX_GRID_POINTS"):
        fns[k] = getattr(cfg, k)
    fns.update(H.make_spies(rec))
    extract("gauNEGF/density.py", ["calcEmin", "integralFit", "integralFitNEGF", "calcFermiBisect",
                                   "calcFermiSecant", "calcFermiMuller", "calcFermiPolyFit"], fns)
    for tag, name, call in H.CASES:
        rec.clear()
        ret = quiet(call, fns[name], H.ProbeG(rec))
        out[f"{tag}_probes"] = H.pack(rec)
        out[f"{tag}_ret"] = H.scalars(ret)
    np.savez_compressed(os.path.join(OUT, "ref_fermi_search.npz"), **out)
    print("ref_fermi_search.npz:", len(out), "arrays;", {t: len(out[f"{t}_probes"]) for t, _, _ in H.CASES})
    print("numpy", np.__version__, "scipy", scipy.__version__)


if __name__ == "__main__":
    sys.exit(main())
