"""Host logic of the adaptive integrations (no GPU): the refinement generator, speculative evaluation of several levels,
integrations that refine together, the batched walk of calcEmin, and bench.py's flop accounting helpers.
Reference: integratePointsAdaptiveANT, gauNEGF/density.py:211-273; calcEmin, :821-836."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from gaunegf_amd import density as D


def _integrand(E, w):
    E = np.asarray(E); w = np.asarray(w)
    return np.array([[np.sum(w / (E - 0.3 + 0.5j)), np.sum(w * np.exp(-E * E))], [0.0, np.sum(w)]], dtype=complex)


MAPS = [lambda x, w: (0.5 * (x + 1) + 0j, 0.5 * w + 0j), lambda x, w: (2.0 * x + 1j, 2.0 * w + 0j)]


def _reference_driver(computePoint, tol, maxN=486):
    """The loop of gauNEGF/density.py:239-273, restated without any speculation (the behaviour to preserve)."""
    prev_x = prev_sumW = P = new_P = None
    N = 2
    while N <= maxN:
        x, w = D.getANTPoints(N)
        if prev_x is None:
            P = computePoint(x[0:2], w[0:2])
        else:
            old = np.isin(np.round(x, 14), np.round(prev_x, 14))
            ratio = float(np.sum(w[old]) / prev_sumW)
            new_P = P * ratio
            new_P += computePoint(x[~old], w[~old])
            maxDP = np.max(np.abs(new_P - P))
            P = new_P.copy()
            if maxDP < tol:
                return new_P
        prev_x = x; prev_sumW = float(np.sum(w)); N *= 3
    return new_P


@pytest.mark.parametrize("tol", [1e-3, 1e-6, 1e-9, 1e-30])
def test_speculative_and_joint_refinement_equal_the_reference_loop(tol):
    """Whatever is evaluated ahead of its turn, and in whatever company, every integration makes the reference's updates
    and stops where the reference stops: with a deterministic integrand all drivers agree BITWISE (tol = 1e-30 never
    converges: the 'reached full grid' exit)."""
    calls = []

    def point(m):
        def f(x, w):
            calls.append(x.size)
            return _integrand(*m(x, w))
        return f
    with contextlib.redirect_stdout(io.StringIO()):
        ref = [_reference_driver(point(m), tol) for m in MAPS]
        n_ref = len(calls)
        plain = [D.integratePointsAdaptiveANT(point(m), tol=tol) for m in MAPS]
        assert len(calls) == 2 * n_ref                                   # same call sequence without computeLevels
        for budget in (0, 6, 64, 512):
            spec = [D.integratePointsAdaptiveANT(point(m), tol=tol, budget=budget,
                                                 computeLevels=lambda nodes, m=m: [_integrand(*m(x, w)) for x, w in nodes])
                    for m in MAPS]
            joint = D.integrateJointlyAdaptiveANT(MAPS, lambda segs: [_integrand(E, w) for E, w in segs], tol=tol, budget=budget)
            for a, b, c, d in zip(ref, plain, spec, joint):
                assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d), (tol, budget)


def _host_refine(requests, tol):
    """negf_gr_int_refine's contract, restated on the host (what Engine.gr_int_refine returns)."""
    out = []
    for E, w, counts, ratios, P in requests:
        conv, maxdps, off = -1, [], 0
        for j, (c, ratio) in enumerate(zip(counts, ratios)):
            val = _integrand(E[off:off + c], w[off:off + c])
            off += c
            if ratio is None:
                assert j == 0 and P is None
                P = val; maxdps.append(np.nan)
                continue
            new_P = P * ratio
            new_P += val
            maxdps.append(np.max(np.abs(new_P - P)))
            P = new_P
            if maxdps[-1] < tol:
                conv = j
                break
        out.append((P, conv, np.array(maxdps + [np.nan] * (len(counts) - len(maxdps)))))
    return out


@pytest.mark.parametrize("tol", [1e-3, 1e-6, 1e-9, 1e-30])
def test_device_side_refinement_driver_equals_the_reference_loop(tol):
    """density._refine_jointly (the refinement itself delegated to negf_gr_int_refine, here restated on the host): the same
    values, the same stopping level and the same messages as the reference loop, whatever the budget -- including
    integrations that need a second round with their running value handed back in."""
    out_ref, out_new = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out_ref):
        ref = [_reference_driver(lambda x, w, m=m: _integrand(*m(x, w)), tol) for m in MAPS]
        D.integrateJointlyAdaptiveANT(MAPS, lambda segs: [_integrand(E, w) for E, w in segs], tol=tol, budget=0)
    rounds = []

    def refine(requests, tol_):
        rounds.append([len(r[2]) for r in requests])
        return _host_refine(requests, tol_)
    for budget in (0, 6, 64, 512):
        with contextlib.redirect_stdout(out_new if budget == 0 else io.StringIO()):
            got = D._refine_jointly(MAPS, refine, tol=tol, budget=budget)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), (tol, budget)
    assert out_new.getvalue() == out_ref.getvalue()
    assert max(len(r) for r in rounds) == 2 and any(r == [6, 6] for r in rounds)     # budget 512: every level in one round


def test_convergence_hints_only_change_what_is_evaluated_ahead():
    """density._refine_jointly with ``hints``: the second of two neighbouring integrations is not sent the levels beyond the
    one its predecessor converged at ahead of its test (and asks for them in a second round if it does need them); values,
    stopping levels and messages are those of the unhinted run."""
    rounds = []

    def refine(requests, tol_):
        rounds.append([sum(r[2]) for r in requests])
        return _host_refine(requests, tol_)
    for tol in (1e-3, 1e-6, 1e-9):
        plain_out, hinted_out = io.StringIO(), io.StringIO()
        with contextlib.redirect_stdout(plain_out):
            plain = D._refine_jointly(MAPS, refine, tol=tol, budget=512)
        hints = [None, None]
        with contextlib.redirect_stdout(io.StringIO()):
            D._refine_jointly(MAPS, refine, tol=tol, budget=512, hints=hints)
        assert all(h is not None and h >= 1 for h in hints)
        del rounds[:]
        with contextlib.redirect_stdout(hinted_out):
            again = D._refine_jointly(MAPS, refine, tol=tol, budget=512, hints=hints)
        for a, b in zip(plain, again):
            assert np.array_equal(a, b)
        assert hinted_out.getvalue() == plain_out.getvalue()
        nodes_to = lambda j: sum(lv[1].size for lv in D._ant_levels(486)[:j + 1])
        assert rounds == [[nodes_to(hints[0]), nodes_to(hints[1])]]                 # one round, exactly the levels needed
    # a hint that is too low costs a second round, not the result
    low = [1, 1]
    del rounds[:]
    with contextlib.redirect_stdout(io.StringIO()):
        got = D._refine_jointly(MAPS, refine, tol=1e-9, budget=512, hints=low)
    for a, b in zip(plain, got):                                                  # (plain: the tol = 1e-9 run above)
        assert np.array_equal(a, b)
    assert len(rounds) == 2 and rounds[0] == [6, 6]


def test_grids_of_the_refined_path_equal_the_level_by_level_grids_bitwise(monkeypatch):
    """density._refine_jointly evaluates an integration's node -> energy map ONCE on the concatenated nodes of the levels it
    requests.  Every map is elementwise: each node must get, bit for bit, the energy and the weight it gets when the levels are
    mapped one by one (the reference's sequence, captured here through a GrInt that never converges)."""
    N = 6
    F = np.diag(np.linspace(-1.0, 1.0, N)); S = np.eye(N)
    per_level = []

    def fake_grint(F_, S_, g_, E, w):
        per_level.append((np.array(E), np.array(w)))
        return np.full((N, N), float(len(per_level) % 2) * 1e6, dtype=complex)      # max|dP| stays huge: every level is visited
    monkeypatch.setattr(D, "GrInt", fake_grint)
    with contextlib.redirect_stdout(io.StringIO()):
        D.densityComplex(F, S, object(), -3.0, 0.2, tol=1e-6, T=300.0)
    assert [e.size for e, _ in per_level] == [2, 4, 12, 36, 108, 324] * 2            # arc, then tail
    monkeypatch.undo()
    got = []

    def fake_refiner(F_, S_, g_):
        def refine(requests, tol):
            got.extend(requests)
            return [(np.zeros((N, N), dtype=complex), -1, np.full(len(r[2]), 1.0)) for r in requests]
        return refine
    monkeypatch.setattr(D._integrate, "GrIntRefiner", fake_refiner)
    monkeypatch.setattr(D, "_speculation_budget", lambda *a: 512)
    with contextlib.redirect_stdout(io.StringIO()):
        D.densityComplex(F, S, object(), -3.0, 0.2, tol=1e-6, T=300.0)
    assert len(got) == 2 and [r[2] for r in got] == [(2, 4, 12, 36, 108, 324)] * 2
    for k, (E, w, counts, ratios, P_in) in enumerate(got):
        off = 0
        for j, c in enumerate(counts):
            e_ref, w_ref = per_level[6 * k + j]
            assert np.array_equal(E[off:off + c], e_ref) and np.array_equal(w[off:off + c], w_ref), (k, j)
            off += c
        assert ratios[0] is None and all(r is not None for r in ratios[1:]) and P_in is None


def test_joint_refinement_batches_requests():
    """Two integrations that converge at different levels: every round is ONE call carrying the requests of those still
    refining; the one that has converged asks for nothing more."""
    rounds = []

    def segments(segs):
        rounds.append([E.size for E, _ in segs])
        return [_integrand(E, w) for E, w in segs]
    smooth = lambda x, w: (0.05 * x + 3.0 + 0j, 0.05 * w + 0j)            # converges at once
    with contextlib.redirect_stdout(io.StringIO()):
        D.integrateJointlyAdaptiveANT([smooth, MAPS[1]], segments, tol=1e-9, budget=6)
    assert rounds[0] == [2, 4, 2, 4]                                       # both ask for levels 2 and 6 (budget 6 nodes)
    assert all(len(r) <= 2 for r in rounds[1:]) and len(rounds[-1]) == 1    # later: one level each, finally only one left
    assert sum(len(r) for r in rounds) < 2 * 6 + 2


def test_levels_are_memoised_and_read_only():
    a = D._ant_levels(486); b = D._ant_levels(486)
    assert a is b and [lv[0] for lv in a] == [2, 6, 18, 54, 162, 486]
    assert [lv[1].size for lv in a] == [2, 4, 12, 36, 108, 324]
    with pytest.raises(ValueError):
        a[1][1][0] = 0.0


def test_speculation_budget_by_size():
    assert D._speculation_budget(np.zeros((60, 60))) == D.SPECULATIVE_POINTS_SMALL
    assert D._speculation_budget(np.zeros((200, 200))) == D.SPECULATIVE_POINTS_ONE_CU
    assert D._speculation_budget(np.zeros((800, 800))) == D.SPECULATIVE_POINTS
    old = D.SPECULATIVE_POINTS
    D.SPECULATIVE_POINTS = 0
    try:
        assert D._speculation_budget(np.zeros((60, 60))) == 0
    finally:
        D.SPECULATIVE_POINTS = old


def test_bench_flop_accounting():
    sys.path.insert(0, ROOT)
    import bench
    per = bench.chain_mfma_flops_per_sweep(50)
    assert 0.80 < per / (24 * 50 ** 3) < 0.92            # 3M + strips + padding; the PMC counter says 0.883
    assert abs(bench.chain_mfma_flops_per_sweep(64) / (24 * 64 ** 3) - 0.75) < 1e-12   # four full tiles, pure 3M
    r = bench.roofline_pair(7.6e12, 5.7e12, 0.1)
    assert abs(r["achieved"] - 76.0) < 1e-9 and abs(r["mfma_executed"] - 57.0) < 1e-9 and r["frac"] < r["frac_algorithmic"] <= 1
    r = bench.roofline_pair(8e12, 6e12, 0.1)             # a 3M product above 3/4 of the peak: a rate, no fraction above 1
    assert r["frac_algorithmic"] is None and abs(r["reference_equivalent_tflops"] - 80.0) < 1e-9 and r["frac"] <= 1
    with pytest.raises(AssertionError):
        bench.roofline_pair(8e12, 9e12, 0.1)             # more than the peak issued: the accounting is wrong
    with pytest.raises(AssertionError):
        bench.roofline_pair(11e12, 6e12, 0.1)            # more than 4/3 of the peak "algorithmic": wrong as well
    assert bench.hbm_fraction(None, 1.0) is None and abs(bench.hbm_fraction(4e12 * 0.5, 0.5) - 0.5) < 1e-12
    with pytest.raises(AssertionError):
        bench.hbm_fraction(9e12, 1.0)


def test_host_blas_thread_limit_is_scoped(monkeypatch):
    """The density step limits the threads of its host BLAS / LAPACK calls (_hostblas.py) and gives the setting back:
    inside ``limited()`` every BLAS pool runs at most host_threads() threads, afterwards what it had before."""
    import importlib
    tpc = pytest.importorskip("threadpoolctl")
    from gaunegf_amd import _hostblas
    monkeypatch.setenv("NEGF_HOST_BLAS_THREADS", "2")
    hb = importlib.reload(_hostblas)
    try:
        assert hb.host_threads() == 2
        before = {d["filepath"]: d["num_threads"] for d in tpc.threadpool_info()}
        with hb.limited():
            assert all(d["num_threads"] <= 2 for d in tpc.threadpool_info())
        assert {d["filepath"]: d["num_threads"] for d in tpc.threadpool_info()} == before

        @hb.limited_call
        def inside():
            return max(d["num_threads"] for d in tpc.threadpool_info())
        import numpy as np
        np.linalg.eigh(np.eye(4))                       # (make sure a BLAS is loaded)
        assert inside() <= 2
        monkeypatch.setenv("NEGF_HOST_BLAS_THREADS", "0")
        hb = importlib.reload(_hostblas)
        assert hb.host_threads() is None
    finally:
        monkeypatch.delenv("NEGF_HOST_BLAS_THREADS", raising=False)
        importlib.reload(_hostblas)


def test_lowest_orbital_energy_routes():
    """calcEmin starts 5 eV below min Re eig(inv(S) F) (density.py:822).  Small or non-Hermitian systems take the
    reference's non-symmetric eigenproblem, bit for bit; a Hermitian system of >= 256 orbitals takes the lowest
    generalised eigenvalue of (F, S) -- the same number to rounding."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import random_system
    F, S = random_system(120, 3)
    assert D._lowest_orbital_energy(F, S) == min(D._orbital_energies(F, S))
    F, S = random_system(300, 4)
    ref = min(D._orbital_energies(F, S))
    assert abs(D._lowest_orbital_energy(F, S) - ref) < 1e-11 * abs(ref)
    Fn = F.copy(); Fn[0, 1] += 1e-3                              # not Hermitian any more: the reference's route
    assert D._lowest_orbital_energy(Fn, S) == min(D._orbital_energies(Fn, S))
    Sbad = S.copy(); Sbad[np.arange(300), np.arange(300)] -= 10.0   # Hermitian but not positive definite: falls back
    assert D._lowest_orbital_energy(F, Sbad) == min(D._orbital_energies(F, Sbad))


def test_calc_emin_reference_route_is_bit_identical(monkeypatch):
    """NEGF_CALC_EMIN_ROUTE=reference (or density.CALC_EMIN_ROUTE): a Hermitian system of 300 orbitals takes the
    reference's route -- min Re eig(inv(S) F), gauNEGF/density.py:822 -- so that Emin and every grid derived from it are
    the reference's bit for bit: the contour nodes / weights densityComplexN hands to GrInt (captured by a spy, the
    golden bookkeeping's method) for the walked Emin are array_equal to those built on the directly computed start
    value, while the default route differs in the last bits (same grids to 1e-12)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import random_system
    F, S = random_system(300, 11)
    ref_start = min(np.sort(np.real(np.linalg.eigvals(np.linalg.solve(S, F)))))

    class Flat:                                     # DOS below tol everywhere: calcEmin returns its start value - 5
        def sigmaTot(self, E): return -1e3j * np.eye(300)

    captured = []

    def spy(F_, S_, g_, E, w):
        captured.append((np.array(E), np.array(w)))
        return np.zeros_like(F_, dtype=complex)

    def grids(emin):
        captured.clear()
        monkeypatch.setattr(D, "GrInt", spy)
        D.densityComplexN(F, S, Flat(), emin, -2.0, 24, T=300)
        return [(e.copy(), w.copy()) for e, w in captured]

    monkeypatch.setattr(D, "_compute_dos_at_energy", lambda E, F_, S_, sig: 0.0)      # (the product's serves it from the GPU)
    monkeypatch.setenv("NEGF_CALC_EMIN_ROUTE", "reference")
    assert D._lowest_orbital_energy(F, S) == ref_start
    emin_ref = D.calcEmin(F, S, Flat(), tol=1e-3)
    assert emin_ref == ref_start - 5
    g_ref, g_direct = grids(emin_ref), grids(ref_start - 5)
    assert len(g_ref) == len(g_direct) > 0
    for (e1, w1), (e2, w2) in zip(g_ref, g_direct):
        assert np.array_equal(e1, e2) and np.array_equal(w1, w2)
    monkeypatch.setattr(D, "CALC_EMIN_ROUTE", "fast")          # the module attribute wins over the environment
    emin_fast = D.calcEmin(F, S, Flat(), tol=1e-3)
    assert abs(emin_fast - emin_ref) < 1e-11 * abs(emin_ref)
    for (e1, w1), (e2, w2) in zip(grids(emin_fast), g_ref):
        assert np.allclose(e1, e2, rtol=1e-12, atol=0) and np.allclose(w1, w2, rtol=1e-12, atol=1e-300)
    monkeypatch.setattr(D, "CALC_EMIN_ROUTE", "nonsense")
    with pytest.raises(ValueError):
        D._lowest_orbital_energy(F, S)


def test_speculation_only_where_levels_are_fused():
    """density._speculation_budget: levels are evaluated ahead of the convergence test only when GrIntSegments really
    fuses them into one pass (a device-lowerable provider -- also per rank of an energy-sharded run, where the fused levels
    share ONE all-reduce, and per spin block); a foreign provider gets the reference's level-by-level sequence
    (budget 0) -- speculated levels would cost a launch each."""
    F = np.eye(60); S = np.eye(60)

    class Foreign:
        def sigmaTot(self, E): return np.zeros((60, 60), dtype=complex)

    class Lowerable(Foreign):
        def _negf_lower(self, eng): raise AssertionError("not evaluated here")
        def sigma(self, E, i): return np.zeros((60, 60), dtype=complex)

    assert D._speculation_budget(F) == D.SPECULATIVE_POINTS_SMALL
    assert D._speculation_budget(F, S, Foreign()) == 0
    assert D._speculation_budget(F, S, Lowerable()) == D.SPECULATIVE_POINTS_SMALL
    from gaunegf_amd import distributed as dist
    import unittest.mock as um
    with um.patch.object(dist, "is_active", lambda: True):
        assert D._speculation_budget(F, S, Lowerable()) == D.SPECULATIVE_POINTS_SMALL
        assert D._speculation_budget(F, S, Foreign()) == 0
    # a block-diagonal spin system: fused when the provider serves its blocks on the device (surfGTest does)
    from gaunegf_amd.surfGTester import surfGTest
    rng = np.random.default_rng(5)
    A = rng.standard_normal((20, 20)); A = A + A.T
    F2 = np.kron(np.eye(2), A); S2 = np.eye(40)
    g = surfGTest(F2, S2, [[0, 20], [19, 39]], -0.1j)
    assert D._speculation_budget(F2, S2, g) == D.SPECULATIVE_POINTS_SMALL
    assert D._speculation_budget(F2, S2, Foreign()) == 0


def test_engine_keeps_conversions_of_unchanged_system_matrices():
    """Engine._c128_cached: the same real array with the same content gets its complex conversion back (same object);
    a change in place, another array or a complex contiguous input are converted / passed through afresh."""
    from gaunegf_amd.engine import Engine
    eng = object.__new__(Engine)                         # (no context: only the host-side helper is exercised)
    rng = np.random.default_rng(1)
    A = rng.standard_normal((160, 160))
    c1 = eng._c128_cached(A)
    assert c1.dtype == np.complex128 and np.array_equal(c1, A)
    assert eng._c128_cached(A) is c1
    A[3, 4] += 1.0                                       # changed in place: a new conversion with the new content
    c2 = eng._c128_cached(A)
    assert c2 is not c1 and c2[3, 4] == A[3, 4]
    B = A.copy()
    assert eng._c128_cached(B) is not c2 and eng._c128_cached(A) is c2
    Z = A.astype(np.complex128)
    assert eng._c128_cached(Z) is Z
    for k in range(6):                                   # the cache holds four matrices
        eng._c128_cached(rng.standard_normal((130, 130)))
    assert len(eng._sys_conv) == 4
    # serial numbers (negf_set_system_keyed): one per kept conversion, the same while the content is, never reused;
    # none for arrays the engine holds no private copy of
    _, k1 = eng._c128_keyed(A)
    assert k1 > 0 and eng._c128_keyed(A)[1] == k1
    A[0, 0] -= 2.0
    _, k2 = eng._c128_keyed(A)
    assert k2 > k1
    assert eng._c128_keyed(Z)[1] == 0                    # the caller's own complex array: may change behind our back
    Zf = Z.copy(); Zf.setflags(write=False)
    zc, kz = eng._c128_keyed(Zf)
    assert zc is Zf and kz > k2 and eng._c128_keyed(Zf)[1] == kz      # frozen: passed through, numbered
    assert eng._c128_keyed(rng.standard_normal((8, 8)))[1] == 0      # small: not kept
