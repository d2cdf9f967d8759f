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
    r = bench.roofline_pair(8e12, 6e12, 0.1)
    assert abs(r["achieved"] - 80.0) < 1e-9 and abs(r["mfma_executed"] - 60.0) < 1e-9 and r["frac"] < r["frac_algorithmic"]
    with pytest.raises(AssertionError):
        bench.roofline_pair(8e12, 9e12, 0.1)             # more than the peak issued: the accounting is wrong


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
