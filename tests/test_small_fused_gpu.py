"""
Systems of n <= 96 orbitals on the GPU: the single-kernel path (assemble + Gauss-Jordan inverse + weighted sum with the
matrix in registers, gaunegf_amd/csrc/k_small_fused.hip) against the numpy oracle and against the kernel sequence larger
systems use (negf_set_small_algo(1)).  Reference: _gr_matrix_ops / _GInt, gauNEGF/integrate.py:67-142; utils.inv,
utils.py:52-54.  Tolerance: 1e-8 relative Frobenius (observed ~1e-14).
"""
import warnings

import numpy as np
import pytest

import oracle
from helpers import MockSigma, chain_lead, const_sigma_pair, random_system, rel_fro

pytestmark = pytest.mark.gpu
TOL = 1e-8


def _const(N, seed):
    from gaunegf_amd.surfGTester import surfGTest
    F, S = random_system(N, seed)
    nc = max(1, N // 10)
    inds, _, _ = const_sigma_pair(N, S, nc, 0.1)
    return F, S, surfGTest(F, S, inds, -0.1j), oracle.ConstSigma(F, S, inds, -0.1j)


@pytest.mark.parametrize("N", [1, 2, 15, 16, 17, 31, 32, 33, 48, 60, 64, 65, 80, 81, 95, 96])
def test_small_fused_G_and_integrals(engine, N):
    """Every tile class (T = ceil(N / 16) = 1 ... 6), sizes on and next to the tile edges: G(E) itself (STORE mode),
    GrInt (ACCUMULATE mode), GrLessInt and the DOS (STORE mode feeding the product / trace kernels), on the real axis
    and off it, against the oracle; and the fused path against the kernel sequence."""
    from gaunegf_amd.integrate import GrBatch, GrInt, GrLessInt
    F, S, g, g_ref = _const(N, 40 + N)
    E = np.concatenate([np.linspace(-2.5, 2.5, 9) + 1e-6j, [0.3 + 0.7j, -1.0 + 0.05j]])
    w = np.linspace(0.5, 1.5, E.size) * (1 + 0.2j)
    G = GrBatch(F, S, g, E)
    for k, e in enumerate(E):
        assert rel_fro(G[k], oracle.gr_point(g_ref.sigmaTot(e), e, F, S)) < TOL, (N, k)
    a = GrInt(F, S, g, E, w)
    assert rel_fro(a, oracle.GrInt(F, S, g_ref, E, w)) < TOL
    b = GrLessInt(F, S, g, E, w, -1)
    assert rel_fro(b, oracle.GrLessInt(F, S, g_ref, E, w, -1)) < TOL
    engine.set_system(F, S)
    tot = engine.dos(g._negf_lower(engine), E, per_site=False)
    for k, e in enumerate(E):
        assert abs(tot[k] - oracle.dos_at_energy(e, F, S, g_ref.sigmaTot(e))) < 1e-8 * max(1.0, abs(tot[k]))
    assert np.array_equal(a, GrInt(F, S, g, E, w))                # bitwise reproducible
    engine.set_small_algo(1)
    try:
        a1 = GrInt(F, S, g, E, w); G1 = GrBatch(F, S, g, E)
    finally:
        engine.set_small_algo(0)
    assert rel_fro(a, a1) < 1e-12 and rel_fro(G, G1) < 1e-12


@pytest.mark.parametrize("N", [17, 60, 96])
def test_small_fused_core_orbital_pivots(engine, N):
    """E S - F in eV has diagonal entries of 1e3 ... 1e5 on core orbitals.  The columns-per-wave kernel scales its pivot row
    as row_p - (1 - 1/pivot) row_p, which loses eps |pivot| relative in that row: with pivots of up to 1e5 the
    single-kernel path still has to meet the 1e-8 bar against the oracle and agree with the kernel sequence to 1e-9."""
    from gaunegf_amd.integrate import GrBatch
    F, S = random_system(N, 400 + N)
    F = F.copy()
    core = np.arange(0, N, 3)
    F[core, core] -= np.geomspace(1e3, 1e5, core.size)             # deep core levels
    from gaunegf_amd.surfGTester import surfGTest
    inds = [list(range(min(2, N // 2))), list(range(N - min(2, N // 2), N))]
    g_dev = surfGTest(F, S, inds, -0.05j)
    g_ref = oracle.ConstSigma(F, S, inds, -0.05j)
    E = np.concatenate([np.linspace(-3.0, 3.0, 7), [0.3 + 0.5j]])
    G = GrBatch(F, S, g_dev, E)
    ref = oracle.gr_batch(F, S, g_ref, E)
    engine.set_small_algo(1)
    try:
        Gk = GrBatch(F, S, g_dev, E)
    finally:
        engine.set_small_algo(0)
    for k in range(E.size):
        assert rel_fro(G[k], ref[k]) < TOL, (N, k, rel_fro(G[k], ref[k]))
        assert rel_fro(G[k], Gk[k]) < 1e-9, (N, k, rel_fro(G[k], Gk[k]))


def test_small_fused_many_points_and_chunks(engine):
    """More energies than resident workgroups (a workgroup sums several points in its partial record) and more than one
    chunk of 16384 points (chunk sums added in order); split-grid additivity and linearity in the weights."""
    from gaunegf_amd.integrate import GrInt
    F, S, g, g_ref = _const(20, 7)
    rng = np.random.default_rng(3)
    M = 3000
    E = rng.uniform(-3, 3, M) + 1e-3j
    w1 = rng.standard_normal(M); w2 = rng.standard_normal(M)
    a = GrInt(F, S, g, E, w1); b = GrInt(F, S, g, E, w2)
    assert rel_fro(GrInt(F, S, g, E, w1 + 2 * w2), a + 2 * b) < 1e-12
    assert rel_fro(GrInt(F, S, g, E[:1111], w1[:1111]) + GrInt(F, S, g, E[1111:], w1[1111:]), a) < 1e-12
    sub = np.arange(0, M, 37)
    assert rel_fro(GrInt(F, S, g, E[sub], w1[sub]), oracle.GrInt(F, S, g_ref, E[sub], w1[sub])) < TOL
    F4, S4, g4, g4_ref = _const(4, 9)
    M = 40000
    E = rng.uniform(-3, 3, M) + 1e-2j; w = rng.standard_normal(M)
    big = GrInt(F4, S4, g4, E, w)
    parts = sum(GrInt(F4, S4, g4, E[k:k + 10000], w[k:k + 10000]) for k in range(0, M, 10000))
    assert rel_fro(big, parts) < 1e-12
    assert rel_fro(GrInt(F4, S4, g4, E[::400], w[::400]), oracle.GrInt(F4, S4, g4_ref, E[::400], w[::400])) < TOL


@pytest.mark.parametrize("N", [1, 17, 32, 60, 64, 96])
def test_small_fused_singular_and_nan(engine, N):
    """An exactly singular matrix and a NaN column in the middle of a batch: reported through info (1-based column),
    the point NaN-filled, the neighbouring energies untouched -- the behaviour of the blocked kernels."""
    from gaunegf_amd.integrate import GrBatch, GrInt

    class Probe:
        def __init__(self, bad): self.bad = bad
        def sigmaTot(self, E):
            z = np.zeros((N, N), dtype=complex)
            if self.bad == "nan" and abs(E - 1.0) < 1e-12:
                z[:, min(3, N - 1)] = np.nan
            return z
        def sigma(self, E, i): return np.zeros((N, N), dtype=complex)

    S = np.eye(N)
    Fz, _ = random_system(N, 7)
    E = np.array([0.5 + 0.1j, 1.0 + 0j, 2.0 + 0.1j])
    for bad, F in (("singular", np.eye(N)), ("nan", Fz)):
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            G = GrBatch(F, S, Probe(bad), E)
        assert any("singular" in str(r.message) for r in rec), bad
        assert np.all(np.isnan(G[1])), bad
        assert engine.last_info[1] != 0 and engine.last_info[0] == 0 and engine.last_info[2] == 0
        if bad == "singular":
            assert engine.last_info[1] == 1
        for k in (0, 2):
            assert rel_fro(G[k], np.linalg.inv(E[k] * S - F)) < TOL, (bad, k)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            out = GrInt(F, S, Probe(bad), E, np.ones(3))
        # (a block-diagonal F, S pair -- the singular case's F = S = 1 -- is integrated block by block and is exactly
        #  zero between the blocks: the NaN point poisons the diagonal blocks)
        assert np.all(np.isnan(np.diag(out)))
        assert engine.last_info[1] != 0


def test_small_fused_foreign_and_block_providers(engine):
    """The other ways Sigma reaches the fused kernel: a host-evaluated provider (dense Sigma per energy), 1-D chain
    leads (contact blocks subtracted through the position map, two contacts of different size) and a Bethe lattice."""
    from gaunegf_amd.integrate import GrInt, GrLessInt
    from gaunegf_amd.surfG1D import surfG
    N = 30
    F, S = random_system(N, 5)
    rng = np.random.default_rng(2)
    base = 0.05 * (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N)))
    c0 = np.zeros((N, N), dtype=complex); c0[:4, :4] = -0.1j * np.eye(4)
    g = MockSigma(base, [c0])
    E, w = oracle.real_axis_grid(-2, 1, 17, 0.0)
    assert rel_fro(GrInt(F, S, g, E, w), oracle.GrInt(F, S, g, E, w)) < TOL
    assert rel_fro(GrLessInt(F, S, g, E, w, 0), oracle.GrLessInt(F, S, g, E, w, 0)) < TOL
    # chain leads of 7 and 5 orbitals on a 40-orbital device
    N = 40
    F, S = random_system(N, 6)
    aL = chain_lead(7, 61); aR = chain_lead(5, 62)
    inds = [list(range(7)), list(range(N - 5, N))]
    kw = dict(taus=[aL[2].copy(), aR[2].copy()], staus=[aL[3].copy(), aR[3].copy()], alphas=[aL[0], aR[0]],
              aOverlaps=[aL[1], aR[1]], betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=1e-3)
    gd = surfG(F, S, inds, **kw); gd.force_iters = 30
    gr = oracle.Chain1DSigma(F, S, inds, kw["taus"], kw["staus"], kw["alphas"], kw["aOverlaps"], kw["betas"], kw["bOverlaps"],
                             eta=1e-3); gr.force_iters = 30
    E, w = oracle.bias_window_grid(-0.3, 0.3, 12, 300.0)
    assert rel_fro(GrInt(F, S, gd, E, w), oracle.GrInt(F, S, gr, E, w)) < TOL
    for ind in (None, 0, -1):
        assert rel_fro(GrLessInt(F, S, gd, E, w, ind), oracle.GrLessInt(F, S, gr, E, w, ind)) < TOL
    engine.set_small_algo(1)
    try:
        seq = GrInt(F, S, gd, E, w)
    finally:
        engine.set_small_algo(0)
    assert rel_fro(GrInt(F, S, gd, E, w), seq) < 1e-12


@pytest.mark.parametrize("N", [12, 60, 96, 130, 300])
def test_segmented_integrals_equal_separate_calls(engine, N):
    """negf_gr_int_seg: several GrInt integrals of one system from one pass (the levels an adaptive integration is about
    to visit) -- every segment's sum equals the integral on that segment alone (up to summation order), through the
    fused small-system path (N <= 96) and through the kernel sequence; an empty segment gives zeros."""
    from gaunegf_amd.integrate import GrInt, GrIntSegments
    F, S, g, g_ref = _const(N, 90 + N)
    rng = np.random.default_rng(N)
    sizes = [2, 4, 12, 0, 36, 7]
    segs = [(rng.uniform(-2, 2, k) + 1j * rng.uniform(0.01, 1.0, k), rng.standard_normal(k) + 1j * rng.standard_normal(k))
            for k in sizes]
    got = GrIntSegments(F, S, g, segs)
    assert len(got) == len(segs)
    for (E, w), a in zip(segs, got):
        if E.size == 0:
            assert not np.any(a)
            continue
        assert rel_fro(a, GrInt(F, S, g, E, w)) < 1e-12
    assert rel_fro(got[2], oracle.GrInt(F, S, g_ref, *segs[2])) < TOL
    again = GrIntSegments(F, S, g, segs)
    assert all(np.array_equal(a, b) for a, b in zip(got, again))


def test_adaptive_integrations_with_and_without_speculation(engine, capsys):
    """densityComplex / densityReal evaluate the levels they are about to visit together (density.SPECULATIVE_POINTS);
    level by level (the reference's call sequence) they must return the same density to rounding, and both equal the
    oracle-served run of the same driver."""
    from gaunegf_amd import density as D
    F, S, g, g_ref = _const(24, 5)
    args = (F, S, g, -6.0, 0.2)
    spec = D.densityComplex(*args, tol=1e-6, T=300.0), D.densityReal(F, S, g, -40.0, -6.0, tol=1e-6, T=0)
    old = D.SPECULATIVE_POINTS
    D.SPECULATIVE_POINTS = 0
    try:
        plain = D.densityComplex(*args, tol=1e-6, T=300.0), D.densityReal(F, S, g, -40.0, -6.0, tol=1e-6, T=0)
    finally:
        D.SPECULATIVE_POINTS = old
    for a, b in zip(spec, plain):
        assert rel_fro(a, b) < 1e-12
    saved = D.GrInt
    D.GrInt = oracle.GrInt
    try:
        ref = D.densityComplex(F, S, g_ref, -6.0, 0.2, tol=1e-6, T=300.0), D.densityReal(F, S, g_ref, -40.0, -6.0, tol=1e-6, T=0)
    finally:
        D.GrInt = saved
    for a, b in zip(spec, ref):
        assert rel_fro(a, b) < TOL


def test_segment_and_cache_argument_checks(engine):
    """Error behaviour of the round-4 entry points: malformed segment tables are NEGF_EINVAL, not a launch; the cache
    knobs reject negative values; the byte budget evicts like the entry count does."""
    import ctypes as C
    from gaunegf_amd._lib import NegfError, NEGF_EINVAL
    F, S, g, _ = _const(10, 3)
    engine.set_system(F, S)
    h = g._negf_lower(engine)
    E = np.ascontiguousarray(np.linspace(-1, 1, 6) + 0.1j); w = np.ascontiguousarray(np.ones(6, dtype=complex))
    out = np.zeros((2, 10, 10), dtype=complex)
    lib, ctx = engine._lib, engine._ctx
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for ends in ([4, 5], [4, 3], [7, 6], [-1, 6]):            # last != m, decreasing, beyond m, negative
        se = np.ascontiguousarray(ends, dtype=np.int32)
        assert lib.negf_gr_int_seg(ctx, h, 6, vp(E), vp(w), 2, vp(se), vp(out), None) == NEGF_EINVAL, ends
    se = np.ascontiguousarray([2, 6], dtype=np.int32)
    assert lib.negf_gr_int_seg(ctx, h, 6, vp(E), vp(w), 0, vp(se), vp(out), None) == NEGF_EINVAL
    assert lib.negf_gr_int_seg(ctx, h, 6, vp(E), vp(w), 2, vp(se), vp(out), None) == 0
    with pytest.raises(NegfError):
        engine.set_chain_cache(-1)
    with pytest.raises(NegfError):
        engine.set_chain_cache(max_bytes=-5)
    with pytest.raises(NegfError):
        engine.set_small_algo(7)


def test_tile_layout_kernel_in_a_fresh_process():
    """NEGF_SMALL_KERNEL=tile selects the first layout of the small-system kernel (16 x 16 thread grid, two barriers per
    pivot step; the default is columns-per-wave with one): kept as an A/B and a cross-check, so it must stay correct."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import oracle
        from test_small_fused_gpu import _const
        from helpers import rel_fro
        from gaunegf_amd.integrate import GrBatch, GrInt
        for N in (1, 17, 60, 81, 96):
            F, S, g, g_ref = _const(N, 40 + N)
            E = np.concatenate([np.linspace(-2.5, 2.5, 5) + 1e-6j, [0.3 + 0.7j]]); w = np.linspace(0.5, 1.5, E.size) * (1 + 0.2j)
            G = GrBatch(F, S, g, E)
            for k, e in enumerate(E):
                assert rel_fro(G[k], oracle.gr_point(g_ref.sigmaTot(e), e, F, S)) < 1e-8, (N, k)
            assert rel_fro(GrInt(F, S, g, E, w), oracle.GrInt(F, S, g_ref, E, w)) < 1e-8
        print("ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NEGF_SMALL_KERNEL="tile")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_panel_form_equals_the_per_step_form_bitwise(engine, tmp_path):
    """NEGF_SMALL_KERNEL=panel factors a wave's columns as one panel (one workgroup barrier per panel; measured slower, kept
    as a cross-check); the default is the same layout with one barrier per pivot step.  Both apply the same operations to every
    element in the same order: G(E) and the weighted sums must agree BITWISE, for every tile class, off the real axis,
    with a singular energy in the batch (same info, same NaN fill)."""
    import os, subprocess, sys, textwrap
    from gaunegf_amd.integrate import GrBatch, GrInt
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from test_small_fused_gpu import _const
        from gaunegf_amd.integrate import GrBatch, GrInt
        out = {}
        for N in (1, 7, 16, 17, 33, 48, 60, 64, 65, 80, 81, 96):
            F, S, g, g_ref = _const(N, 40 + N)
            E = np.concatenate([np.linspace(-2.5, 2.5, 5) + 1e-6j, [0.3 + 0.7j]]); w = np.linspace(0.5, 1.5, E.size) * (1 + 0.2j)
            out["G%%d" %% N] = GrBatch(F, S, g, E); out["I%%d" %% N] = GrInt(F, S, g, E, w)
        np.savez(sys.argv[1], **out)
        print("ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    f = str(tmp_path / "panel.npz")
    r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, NEGF_SMALL_KERNEL="panel"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    ref = np.load(f)
    for N in (1, 7, 16, 17, 33, 48, 60, 64, 65, 80, 81, 96):
        F, S, g, g_ref = _const(N, 40 + N)
        E = np.concatenate([np.linspace(-2.5, 2.5, 5) + 1e-6j, [0.3 + 0.7j]]); w = np.linspace(0.5, 1.5, E.size) * (1 + 0.2j)
        assert np.array_equal(GrBatch(F, S, g, E), ref["G%d" % N]), N
        assert np.array_equal(GrInt(F, S, g, E, w), ref["I%d" % N]), N


def test_segments_across_workspace_chunks(engine):
    """The kernel-sequence path of negf_gr_int_seg streams the energies through the workspace in chunks: segments that
    straddle chunk boundaries (a workspace of 5 energies, segments of 2, 4, 12 and 7 points) still sum correctly."""
    from gaunegf_amd.integrate import GrInt, GrIntSegments
    F, S, g, _ = _const(130, 77)
    rng = np.random.default_rng(5)
    segs = [(rng.uniform(-2, 2, k) + 0.05j, rng.standard_normal(k) + 0j) for k in (2, 4, 12, 7)]
    ref = [GrInt(F, S, g, E, w) for E, w in segs]
    engine.set_batch(5)
    try:
        got = GrIntSegments(F, S, g, segs)
    finally:
        engine.set_batch(0)
    for a, b in zip(got, ref):
        assert rel_fro(a, b) < 1e-12


def test_segments_on_the_windowed_inverse_and_on_spin_blocks(engine):
    """negf_gr_int_seg above the single-workgroup sizes: every segment's sum reads the windowed inverse through its pivot
    bookkeeping (no gather), segments straddling workspace chunks included; an exactly singular energy poisons its own
    segment only.  A block-diagonal spin system goes block by block, every block's segments in one pass."""
    import warnings
    from gaunegf_amd.integrate import GrInt, GrIntSegments, GrLessInt, GrLessIntSegments
    from gaunegf_amd.surfGTester import surfGTest
    F, S, g, g_ref = _const(300, 41)
    rng = np.random.default_rng(9)
    segs = [(rng.uniform(-2, 2, k) + 0.05j, rng.standard_normal(k) + 0j) for k in (3, 1, 20, 9)]
    ref = [GrInt(F, S, g, E, w) for E, w in segs]
    for batch in (0, 7):
        engine.set_batch(batch)
        try:
            got = GrIntSegments(F, S, g, segs)
        finally:
            engine.set_batch(0)
        for a, b in zip(got, ref):
            assert rel_fro(a, b) < 1e-12
    assert rel_fro(got[2], oracle.GrInt(F, S, g_ref, *segs[2])) < TOL
    # E S - F - Sigma exactly singular at one energy of segment 1 (F = S, Sigma = 0 staged per energy: E = 1 gives the zero matrix)
    engine.set_system(S, S)
    Es = [np.array([0.5 + 0.1j, 2.0 + 0.1j]), np.array([1.0 + 0j, 3.0 + 0.2j])]
    h = engine.sigma_precomputed(np.zeros((4, 300, 300), dtype=complex))
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            out = engine.gr_int_seg(h, [(e, np.ones(2) + 0j) for e in Es])
    finally:
        engine.sigma_free(h)
    Si = np.linalg.inv(S)
    assert rel_fro(out[0], Si / (0.5 + 0.1j - 1) + Si / (2.0 + 0.1j - 1)) < TOL and np.all(np.isnan(out[1].real))
    assert list(np.nonzero(engine.last_info)[0]) == [2]
    # spin blocks
    N = 130
    Fa, Sa = random_system(N, 3); Fb, _ = random_system(N, 4)
    F2 = np.block([[Fa, np.zeros((N, N))], [np.zeros((N, N)), Fb]]); S2 = np.kron(np.eye(2), Sa)
    inds = [[0, 1, 2, N, N + 1, N + 2], [N - 3, N - 2, N - 1, 2 * N - 3, 2 * N - 2, 2 * N - 1]]
    g2 = surfGTest(F2, S2, inds, -0.1j)
    segs2 = [(rng.uniform(-1, 1, k) + 0.02j, rng.standard_normal(k) + 0j) for k in (2, 11)]
    for a, (E, w) in zip(GrIntSegments(F2, S2, g2, segs2), segs2):
        b = GrInt(F2, S2, g2, E, w)
        assert rel_fro(a, b) < 1e-12 and not np.any(a[:N, N:]) and not np.any(a[N:, :N])
    for a, (E, w) in zip(GrLessIntSegments(F2, S2, g2, segs2, -1), segs2):
        assert rel_fro(a, GrLessInt(F2, S2, g2, E, w, -1)) < 1e-12


@pytest.mark.parametrize("N", [24, 60, 130, 300, 600])       # (600: above 512 orbitals a launch per level instead of one workgroup per integration)
def test_refinement_on_the_device_equals_the_host_refinement(engine, N, capsys):
    """negf_gr_int_refine: the nested-rule update and stopping test of density.py:239-268 run by the library on the level sums
    it has just computed.  Against the host refinement of the same sums (Engine.gr_int_seg + numpy): the same value BITWISE
    at the same level, the same maxDP per level to an ulp of hypot; an integration that does not converge in its first call
    continues from its running value; densityComplex gives the same density and the same messages with the switch on and
    off, and agrees with the oracle-served integration."""
    from gaunegf_amd import density as D
    F, S, g, g_ref = _const(N, 300 + N)
    engine.set_system(F, S)
    h = g._negf_lower(engine)
    levels = D._ant_levels(486)
    half_width, mid, radius = D._contour(-3.0, 0.2, 300.0)
    arc = lambda x, w: (mid + radius * np.exp(1j * (np.pi / 2 * (x + 1))), np.pi / 2 * w * (1j * radius * np.exp(1j * (np.pi / 2 * (x + 1)))))
    tail = lambda x, w: (half_width * x + 0.2 + 0j, half_width * w + 0j)
    for tol in (1e-3, 1e-6, 1e-30):
        # host refinement on the segment sums
        want = []
        # (ONE pass over both integrations' levels, as the refining call makes it: the same level sums to the bit)
        all_sums = engine.gr_int_seg(h, [grid(lv[1], lv[2]) for grid in (arc, tail) for lv in levels])
        for k in range(2):
            sums = all_sums[6 * k:6 * k + 6]
            P, conv, dps = sums[0].copy(), -1, [np.nan]
            for j in range(1, len(levels)):
                new_P = P * levels[j][3]
                new_P += sums[j]
                dps.append(np.max(np.abs(new_P - P)))
                P = new_P
                if dps[-1] < tol:
                    conv = j
                    break
            want.append((P, conv, dps))
        # one call, all levels, both integrals
        def request(grid, lo, hi, P_in):
            x, w, counts, ratios = D._level_group(486, lo, hi)
            return grid(x, w) + (counts, ratios, P_in)
        got = engine.gr_int_refine(h, [request(grid, 0, 6, None) for grid in (arc, tail)], tol)
        for (P, conv, dps), (Pd, convd, dpsd) in zip(want, got):
            assert convd == conv and np.array_equal(Pd, P)
            assert np.allclose(dpsd[1:len(dps)], dps[1:], rtol=1e-14, atol=0) and np.all(np.isnan(dpsd[len(dps):])) and np.isnan(dpsd[0])
        # two calls: levels 0..2, then the rest from the running value
        first = engine.gr_int_refine(h, [request(arc, 0, 3, None)], tol)[0]
        if first[1] < 0:
            second = engine.gr_int_refine(h, [request(arc, 3, 6, first[0])], tol)[0]
            # (the level sums of another call may be grouped differently in the workspace: rounding, not bits)
            assert rel_fro(second[0], want[0][0]) < 1e-14 and (second[1] + 3 if second[1] >= 0 else -1) == want[0][1]
        else:
            assert first[1] == want[0][1] and rel_fro(first[0], want[0][0]) < 1e-14
    capsys.readouterr()
    dens, text = {}, {}
    for on in (True, False):
        D.REFINE_ON_DEVICE = on
        try:
            dens[on] = D.densityComplex(F, S, g, -3.0, 0.2, tol=1e-6, T=300.0)
        finally:
            D.REFINE_ON_DEVICE = True
        text[on] = capsys.readouterr().out
    assert text[True] == text[False] and "converged" in text[True]
    assert rel_fro(dens[True], dens[False]) < 1e-14
    saved = D.GrInt
    D.GrInt = oracle.GrInt
    try:
        ref = D.densityComplex(F, S, g_ref, -3.0, 0.2, tol=1e-6, T=300.0)
    finally:
        D.GrInt = saved
    assert rel_fro(dens[True], ref) < TOL


def test_fixed_grid_density_step_in_one_pass(engine):
    """density.densityEquilibriumN = (densityRealN, densityComplexN) from one pass over the three grids; NEGFE.FockToP with
    fixed grids at a given Fermi level uses it and reproduces the oracle-served step."""
    import contextlib, io
    from gaunegf_amd import density as D
    from gaunegf_amd.scfE import NEGFE
    F, S, g, g_ref = _const(60, 5)
    with contextlib.redirect_stdout(io.StringIO()):
        calls0 = engine.counters["calls"]
        Pr, Pc = D.densityEquilibriumN(F, S, g, -50.0, -3.0, 0.1, 24, 40, 300.0)
        assert engine.counters["calls"] - calls0 == 1
        assert rel_fro(Pr, D.densityRealN(F, S, g, -50.0, -3.0, 24, T=0, showText=False)) < 1e-12
        assert rel_fro(Pc, D.densityComplexN(F, S, g, -3.0, 0.1, 40, 300.0, showText=False)) < 1e-12

        def step(gobj):
            n = NEGFE(F, S, gobj, ne=24, spin='r', T=300.0, Eminf=-50.0)
            n.setIntegralLimits(N1=40, N2=24, Emin=-3.0)
            n.setVoltage(0.0, fermi=0.1)
            n.FockToP()
            return n.P
        calls0 = engine.counters["calls"]
        P = step(g)
        assert engine.counters["calls"] - calls0 == 1
        saved = (D.GrInt, D.GrLessInt)
        D.GrInt, D.GrLessInt = oracle.GrInt, oracle.GrLessInt
        try:
            P_ref = step(g_ref)
        finally:
            D.GrInt, D.GrLessInt = saved
    assert rel_fro(P, P_ref) < TOL


@pytest.mark.parametrize("N", [24, 60, 130])
def test_segmented_lesser_integrals_and_adaptive_bias_window(engine, N, capsys):
    """negf_gless_int_seg: every segment's G Gamma G^H sum equals GrLessInt on that segment alone (ind = None, 0, -1;
    through the compact and the dense coupling products, a workspace smaller than the grid); densityGrid with its levels
    evaluated together equals level by level and the oracle-served run."""
    from gaunegf_amd import density as D
    from gaunegf_amd.integrate import GrLessInt, GrLessIntSegments
    F, S, g, g_ref = _const(N, 200 + N)
    rng = np.random.default_rng(N)
    segs = [(rng.uniform(-1, 1, k) + 0j, rng.standard_normal(k) + 0j) for k in (2, 4, 12, 5)]
    for ind in (None, 0, -1):
        got = GrLessIntSegments(F, S, g, segs, ind)
        for (E, w), a in zip(segs, got):
            assert rel_fro(a, GrLessInt(F, S, g, E, w, ind)) < 1e-12, ind
        assert rel_fro(got[2], oracle.GrLessInt(F, S, g_ref, *segs[2], ind)) < TOL
    engine.set_gamma_algo(1); engine.set_batch(7)
    try:
        dense = GrLessIntSegments(F, S, g, segs, -1)
    finally:
        engine.set_gamma_algo(0); engine.set_batch(0)
    for a, (E, w) in zip(dense, segs):
        assert rel_fro(a, GrLessInt(F, S, g, E, w, -1)) < 1e-12
    spec = D.densityGrid(F, S, g, 0.25, -0.25, ind=-1, tol=1e-7, T=300.0)
    old = D.SPECULATIVE_POINTS
    D.SPECULATIVE_POINTS = 0
    try:
        plain = D.densityGrid(F, S, g, 0.25, -0.25, ind=-1, tol=1e-7, T=300.0)
    finally:
        D.SPECULATIVE_POINTS = old
    assert rel_fro(spec, plain) < 1e-12
    saved = D.GrLessInt
    D.GrLessInt = oracle.GrLessInt
    try:
        ref = D.densityGrid(F, S, g_ref, 0.25, -0.25, ind=-1, tol=1e-7, T=300.0)
    finally:
        D.GrLessInt = saved
    assert rel_fro(spec, ref) < TOL
