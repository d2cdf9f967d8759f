"""
Parity of the HIP engine (through the C ABI) against the numpy oracle and the
committed golden vectors.  Run on the MI355X box:  pytest tests -m gpu

Tolerances (BASELINE.json north_star): grid bookkeeping bit-exact (test_grids.py);
complex128 G(E) and everything derived from it within 1e-8 RELATIVE FROBENIUS of
the reference CPU path (TOL below; observed errors are ~1e-13).  Self-energy
fixed points are compared at a fixed trip count (1e-10) and, free running, within
10*conv (SURVEY.md section 7 "Fixed-point semantics").
"""
import os
import warnings

import numpy as np
import pytest

import oracle
from helpers import MockSigma, chain_lead, const_sigma_pair, random_system, rel_fro

pytestmark = pytest.mark.gpu
TOL = 1e-8


# --------------------------------------------------------------------------- #
def _const_provider(N, seed, nc=None, gamma=0.1):
    from gaunegf_amd.surfGTester import surfGTest
    F, S = random_system(N, seed)
    nc = nc or max(1, N // 10)
    inds, s1, s2 = const_sigma_pair(N, S, nc, gamma)
    g_dev = surfGTest(F, S, inds, -1j * gamma)
    g_ref = oracle.ConstSigma(F, S, inds, -1j * gamma)
    return F, S, g_dev, g_ref


def test_mfma_fragment_layout(engine):
    assert engine.selftest_mfma() == 0.0


def test_library_is_loaded_in_process():
    # the round-end check looks for the in-tree .so among the loaded objects
    maps = open("/proc/self/maps").read()
    from gaunegf_amd.engine import get_engine
    get_engine()
    maps = open("/proc/self/maps").read()
    assert "libnegf_hip.so" in maps


@pytest.mark.parametrize("N,algo", [(n, a) for n in (1, 3, 17, 32, 40, 64, 100, 200, 256) for a in (1, 2)] +
                         [(n, 2) for n in (257, 300, 383, 384, 500, 512, 513, 600, 1030)])
def test_G_of_E_per_energy(engine, N, algo):
    """G(E) = solve(E S - F - Sigma, I) for every energy: unblocked kernel (algo 1), blocked
    MFMA kernels (algo 2: one workgroup per matrix with a 32-column panel up to n=256, two-level windowed above)."""
    from gaunegf_amd.integrate import GrBatch
    F, S, g_dev, g_ref = _const_provider(N, 100 + N)
    E = np.concatenate([np.linspace(-3, 3, 9), np.array([0.3 + 0.5j, -1.2 + 2j, 0.05 + 1e-3j])])
    if N > 256:
        E = E[[0, 4, 10]]
    engine.set_inverse_algo(algo)
    try:
        G = GrBatch(F, S, g_dev, E)
    finally:
        engine.set_inverse_algo(0)
    ref = oracle.gr_batch(F, S, g_ref, E)
    for k in range(len(E)):
        assert rel_fro(G[k], ref[k]) < TOL, (N, algo, k, rel_fro(G[k], ref[k]))


@pytest.mark.parametrize("N,M", [(320, 11), (333, 8), (449, 17)])
def test_windowed_inverse_batch_shapes(engine, N, M):
    """The column-block update of the windowed inverse maps workgroups to (matrix, column block) through an
    XCD-aware order padded to multiples of eight matrices: batches that are not multiples of eight, a dimension
    that is a multiple of the 64-column window (320), a last window of 13 columns (333) and of one column (449)."""
    from gaunegf_amd.integrate import GrBatch
    F, S, g_dev, g_ref = _const_provider(N, 300 + N, nc=20)
    E = np.linspace(-2.5, 2.5, M) + 0.02j
    G = GrBatch(F, S, g_dev, E)
    for k in (0, M // 2, M - 1):
        ref = oracle.gr_batch(F, S, g_ref, E[k:k + 1])[0]
        assert rel_fro(G[k], ref) < TOL, (N, M, k, rel_fro(G[k], ref))
    # every matrix of the batch: residual against its own system
    for k in range(M):
        A = E[k] * S - F - np.asarray(g_ref.sigmaTot(E[k]))
        assert np.linalg.norm(G[k] @ A - np.eye(N)) / np.sqrt(N) < 1e-9, (N, M, k)


def test_blocked_inverses_random_shapes(engine):
    """Seeded random dimensions and batch sizes through the blocked inverses (the round-3 fuzz of scripts/fuzz_inverse.py in
    small): odd window counts, last windows and sub-panels of any width, batches on either side of the eight-wave / lean
    update selection; every sampled matrix checked by its residual."""
    from gaunegf_amd.integrate import GrBatch
    from gaunegf_amd.surfGTester import surfGTest
    rng = np.random.default_rng(2026)
    cases = [(int(rng.integers(257, 700)), int(rng.integers(1, 40))) for _ in range(7)]
    cases += [(int(rng.integers(33, 257)), int(rng.integers(1, 60))) for _ in range(3)] + [(int(rng.integers(1025, 1200)), 2)]
    for n, m in cases:
        F, S = random_system(n, int(rng.integers(1 << 30)))
        nc = max(2, n // 20)
        g = surfGTest(F, S, [list(range(nc)), list(range(n - nc, n))], -0.1j)
        E = np.linspace(-2.0, 2.0, m) + 0.02j
        G = GrBatch(F, S, g, E)
        sig = g.sigmaTot(0.0)
        for k in sorted({0, m // 2, m - 1}):
            A = E[k] * S - F - sig
            assert np.linalg.norm(G[k] @ A - np.eye(n)) / np.sqrt(n) < 1e-9, (n, m, k)


@pytest.mark.parametrize("N,M,algo", [(n, m, a) for a in (3, 4) for n, m in
                                      ((64, 5), (100, 9), (209, 13), (256, 7), (300, 6), (333, 8), (449, 17), (512, 3), (513, 2),
                                       (650, 9), (1024, 2))])
def test_window_kernels_by_force(engine, N, M, algo):
    """Both window-kernel families at every size class whatever the auto rule picks for the batch
    (negf_set_inverse_algo 3: the register-strip kernel of gj_strip.h -- n <= 256 with sub-windows of 32, 257 ... 512 the
    lean two-rows-per-lane form, above the eight-wave form; 4: the team kernels / the single-workgroup kernel): ragged
    last windows (n = 100: 36 columns, 209: 17, 333: 13, 449: 1, 513: 1 -- a last sub-window narrower than a chunk of
    four), rows that end inside a wave, inside a slab, on the slab boundary (256, 512, 1024).  Oracle on a sample,
    the residual on every matrix."""
    from gaunegf_amd.integrate import GrBatch
    F, S, g_dev, g_ref = _const_provider(N, 900 + N, nc=min(20, N // 4))
    E = np.linspace(-2.5, 2.5, M) + 0.02j
    engine.set_inverse_algo(algo)
    try:
        G = GrBatch(F, S, g_dev, E)
    finally:
        engine.set_inverse_algo(0)
    for k in (0, M - 1):
        ref = oracle.gr_batch(F, S, g_ref, E[k:k + 1])[0]
        assert rel_fro(G[k], ref) < TOL, (N, M, algo, k, rel_fro(G[k], ref))
    for k in range(M):
        A = E[k] * S - F - np.asarray(g_ref.sigmaTot(E[k]))
        assert np.linalg.norm(G[k] @ A - np.eye(N)) / np.sqrt(N) < 1e-9, (N, M, algo, k)


@pytest.mark.parametrize("N,algo", [(n, a) for n in (100, 240, 300, 400, 700) for a in (3, 4)])
def test_singular_and_nan_matrices_by_window_kernel(engine, N, algo):
    """The singular / NaN reporting of test_singular_and_nan_matrices_in_blocked_kernels through each window-kernel family."""
    from gaunegf_amd.integrate import GrBatch

    class Probe:
        def __init__(self, bad): self.bad = bad
        def sigmaTot(self, E):
            z = np.zeros((N, N), dtype=complex)
            if self.bad == "nan" and abs(E - 1.0) < 1e-12:
                z[:, 3] = np.nan
            return z
        def sigma(self, E, i): return np.zeros((N, N), dtype=complex)

    S = np.eye(N)
    Fz, _ = random_system(N, 7)
    E = np.array([0.5 + 0.1j, 1.0 + 0j, 2.0 + 0.1j])
    engine.set_inverse_algo(algo)
    try:
        for bad, F in (("singular", np.eye(N)), ("nan", Fz)):
            with warnings.catch_warnings(record=True):
                warnings.simplefilter("always")
                G = GrBatch(F, S, Probe(bad), E)
            assert np.all(np.isnan(G[1])), bad
            assert engine.last_info[1] != 0 and engine.last_info[0] == 0 and engine.last_info[2] == 0
            for k in (0, 2):
                ref = np.linalg.inv(E[k] * S - F)
                assert rel_fro(G[k], ref) < TOL, (bad, k)
    finally:
        engine.set_inverse_algo(0)


@pytest.mark.parametrize("N,M", [(230, 70), (300, 40), (333, 37), (650, 9)])
def test_GrInt_reads_the_windowed_inverse_through_its_permutation(engine, N, M):
    """GrInt on the windowed path never writes G: the inverse leaves its gather out and the weighted sum reads the reduced
    matrices through the pivot bookkeeping (launch_accumulate_perm; same chunks, same order of additions as the
    gather + accumulate sequence).  Against the oracle, against the weighted sum of the G(E) that GrBatch DOES gather
    (1e-13: a host sum in another order), run-to-run bitwise, and with a singular energy in the batch: NaN result,
    info at that energy only."""
    from gaunegf_amd.integrate import GrBatch, GrInt
    F, S, g_dev, g_ref = _const_provider(N, 700 + N, nc=20)
    E = np.linspace(-2.5, 2.5, M) + 0.03j
    w = (np.cos(np.arange(M)) + 1.5) / M * (1 + 0.3j)
    got = GrInt(F, S, g_dev, E, w)
    assert rel_fro(got, oracle.GrInt(F, S, g_ref, E, w)) < TOL
    G = GrBatch(F, S, g_dev, E)
    assert rel_fro(got, np.tensordot(w, G, axes=(0, 0))) < 1e-13
    assert np.array_equal(got, GrInt(F, S, g_dev, E, w))

    class Probe:                          # Sigma = 0 and F = S (a full matrix: no spin-block split): E = 1 makes E S - F == 0
        def sigmaTot(self, E_): return np.zeros((N, N), dtype=complex)
        def sigma(self, E_, i): return np.zeros((N, N), dtype=complex)
    Es = np.array([0.5 + 0.1j, 1.0 + 0j, 2.0 + 0.1j])
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        bad = GrInt(S, S, Probe(), Es, np.ones(3, dtype=complex))
    assert np.all(np.isnan(bad)) and any("singular" in str(r.message) for r in rec)
    assert engine.last_info[1] != 0 and engine.last_info[0] == 0 and engine.last_info[2] == 0


@pytest.mark.parametrize("N,M", [(300, 480), (449, 640), (150, 300), (200, 520)])
def test_windowed_inverse_large_batches(engine, N, M):
    """(N = 150, 200: below the switch-over dimension the windowed path takes the batches of more matrices than the chip
    has CUs -- the C2-sized systems at full batch -- and the single-workgroup kernel the rest.)
    Batches large enough for the throughput configuration of the windowed inverse: four stream groups of >= 120
    matrices, windows in PAIRS (fused update kernel, two lean workgroups per CU) and the lean single-window update
    -- small batches run the windows one by one with eight-wave workgroups.  N = 300: two pairs and an odd last window of
    44 columns; N = 449: four pairs, the last window one column wide.  Checked through the weighted sum over all
    energies (one wrong matrix among M would show at 1/M) and through residuals of a sample."""
    from gaunegf_amd.integrate import GrInt, GrBatch
    F, S, g_dev, g_ref = _const_provider(N, 500 + N, nc=20)
    E = np.linspace(-2.5, 2.5, M) + 0.03j
    w = (np.cos(np.arange(M)) + 1.5) / M + 0.0j
    assert rel_fro(GrInt(F, S, g_dev, E, w), oracle.GrInt(F, S, g_ref, E, w)) < TOL
    G = GrBatch(F, S, g_dev, E)
    for k in (0, 1, M // 3, M // 2, M - 2, M - 1):
        A = E[k] * S - F - np.asarray(g_ref.sigmaTot(E[k]))
        assert np.linalg.norm(G[k] @ A - np.eye(N)) / np.sqrt(N) < 1e-9, (N, M, k)


@pytest.mark.parametrize("N,M", [(12, 12), (60, 100), (200, 64), (150, 6), (330, 5)])
def test_GrInt_GrLessInt_const_sigma(engine, N, M):
    # (G Gamma G^H runs as a Hermitian product: block tiles above the diagonal computed and mirrored -- one block
    #  at N <= 64, the flexible-block kernel at N = 200, 3 x 3 and 6 x 6 blocks of 64 with a ragged edge at 150 / 330)
    from gaunegf_amd.integrate import GrInt, GrLessInt
    F, S, g_dev, g_ref = _const_provider(N, 7 + N)
    E, w = oracle.contour_grid(-4.0, 0.2, M if M % 2 == 0 else M + 1, 300.0)
    assert rel_fro(GrInt(F, S, g_dev, E, w), oracle.GrInt(F, S, g_ref, E, w)) < TOL
    Eg, wg = oracle.bias_window_grid(-0.25, 0.25, M, 300.0)
    for ind in (None, 0, -1, 1):
        got = GrLessInt(F, S, g_dev, Eg, wg, ind)
        assert rel_fro(got, oracle.GrLessInt(F, S, g_ref, Eg, wg, ind)) < TOL, ind


@pytest.mark.parametrize("size", [12, 40])
def test_reference_consistency_vectors(engine, golden_num, size):
    """The reference's own consistency test (tests/test_computation_consistency.py):
    its numpy loops' outputs are the golden values; a foreign provider goes through
    the host-callback path."""
    from gaunegf_amd.integrate import GrInt, GrLessInt
    g = golden_num
    F, S = g[f"cc{size}_F"], g[f"cc{size}_S"]
    prov = MockSigma(g[f"cc{size}_sigma_base"], [g[f"cc{size}_sigma_c0"], g[f"cc{size}_sigma_c1"]])
    E, w = g[f"cc{size}_E"], g[f"cc{size}_w"]
    assert rel_fro(GrInt(F, S, prov, E, w), g[f"cc{size}_gr"]) < TOL
    assert rel_fro(GrLessInt(F, S, prov, E, w), g[f"cc{size}_gless_none"]) < TOL
    assert rel_fro(GrLessInt(F, S, prov, E, w, 0), g[f"cc{size}_gless_0"]) < TOL
    assert rel_fro(GrLessInt(F, S, prov, E, w, 1), g[f"cc{size}_gless_1"]) < TOL
    # the reference grades max-abs error < 1e-10 as "GOOD - consistent" (:226-233)
    assert np.max(np.abs(GrInt(F, S, prov, E, w) - g[f"cc{size}_gr"])) < 1e-10


def test_api_assertions_and_edges(engine):
    from gaunegf_amd.integrate import GrInt, GrLessInt
    F, S, g_dev, g_ref = _const_provider(8, 3)
    with pytest.raises(AssertionError, match="same length"):
        GrInt(F, S, g_dev, np.array([0.1, 0.2]), np.array([1.0]))
    with pytest.raises(AssertionError, match="same shape"):
        GrInt(F, S[:4, :4], g_dev, np.array([0.1]), np.array([1.0]))
    # empty grid (callers pass boolean-masked slices that may be empty, density.py:254)
    out = GrInt(F, S, g_dev, np.array([]), np.array([]))
    assert out.shape == (8, 8) and not np.any(out)
    # a single point, and a workspace batch smaller than the grid (multi-sweep)
    E, w = oracle.real_axis_grid(-3, 0.1, 23, 300.0)
    ref = oracle.GrInt(F, S, g_ref, E, w)
    engine.set_batch(5)
    try:
        assert rel_fro(GrInt(F, S, g_dev, E, w), ref) < TOL
        assert rel_fro(GrLessInt(F, S, g_dev, E, w, -1), oracle.GrLessInt(F, S, g_ref, E, w, -1)) < TOL
    finally:
        engine.set_batch(0)
    assert rel_fro(GrInt(F, S, g_dev, E[:1], w[:1]), oracle.GrInt(F, S, g_ref, E[:1], w[:1])) < TOL
    # invalid contact index
    from gaunegf_amd._lib import NegfError
    with pytest.raises(NegfError):
        GrLessInt(F, S, g_dev, E, w, 5)


def test_workspace_is_stable_across_entry_points(engine):
    """One workflow on one grid -- GrInt, calculate_transmission, GrLessInt, DOS, GrInt -- must not re-allocate
    the batch workspace between calls: the transmission needs two work areas per energy, and once it has run on
    a context the workspace is sized for that (contexts that only integrate keep the single-grid size)."""
    from gaunegf_amd.integrate import GrInt, GrLessInt
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission, calculate_dos
    N = 112                                               # (above 96: GrInt of a smaller system needs no workspace at all)
    F, S, g_dev, g_ref = _const_provider(N, 19)
    E = np.linspace(-1.0, 1.0, 37); w = np.full(37, 2.0 / 37)
    sc = SigmaCalculator(g_dev.sig[0], g_dev.sig[1])
    calculate_transmission(F, S, sc, E[:3])              # from here on the context knows the workflow
    GrInt(F, S, g_dev, E, w)
    b0 = engine.get_batch()
    assert b0 >= 2 * len(E)
    for call in (lambda: calculate_transmission(F, S, sc, E), lambda: GrLessInt(F, S, g_dev, E, w, -1),
                 lambda: calculate_dos(F, S, sc, E), lambda: GrInt(F, S, g_dev, E[:11], w[:11]),
                 lambda: calculate_transmission(F, S, sc, E[:5])):
        call()
        assert engine.get_batch() == b0


def test_explicit_gammas_need_not_be_hermitian(engine):
    """The per-energy kernels take the coupling matrices as arguments (transport.py:150-157): T = Re Tr[g1 G g2 G^H] for
    ANY g1, g2.  The library computes G g2 G^H as a Hermitian product only for coupling matrices it formed itself
    (Gamma = i (Sigma - Sigma^H)); matrices handed in by the caller take the full product."""
    from gaunegf_amd.transport import _transmission_kernel_restricted
    N = 70                                              # two block tiles of 64: an upper and a lower block exist
    F, S = random_system(N, 91)
    rng = np.random.default_rng(5)
    sig = -0.05j * np.eye(N)
    for herm in (True, False):
        g1 = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        g2 = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        if herm:
            g1, g2 = g1 + g1.conj().T, g2 + g2.conj().T
        E = 0.3
        G = np.linalg.inv(E * S - F - sig)
        ref = np.real(np.trace(g1 @ G @ g2 @ G.conj().T))
        got = _transmission_kernel_restricted(E, F, S, sig, g1, g2)
        assert abs(got - ref) <= 1e-9 * max(1.0, abs(ref)), (herm, got, ref)


def test_keyed_resident_systems(engine):
    """Systems of >= 128 x 128 reach the library as the engine's private complex copies and carry a key
    (negf_set_system_keyed): a repeated or alternating system is selected without the bytewise comparison, a third
    system evicts the least recently used one, a matrix changed IN PLACE gets a new copy and a new key, and frozen
    (read-only, owned) arrays are numbered without being copied.  Every call must give the result of its own (F, S)."""
    from gaunegf_amd.integrate import GrInt
    from gaunegf_amd.surfGTester import surfGTest
    N = 160
    systems = [random_system(N, 80 + k) for k in range(3)]
    E = np.linspace(-1.0, 1.0, 7) + 0.05j
    w = np.full(7, 0.25) + 0.0j
    inds = [list(range(5)), list(range(N - 5, N))]

    def check(F, S):
        g = surfGTest(F, S, inds, -0.1j)
        ref = oracle.GrInt(np.asarray(F), np.asarray(S), oracle.ConstSigma(np.asarray(F), np.asarray(S), inds, -0.1j), E, w)
        assert rel_fro(GrInt(F, S, g, E, w), ref) < TOL
    for k in (0, 1, 0, 1, 2, 0, 1, 2, 2):
        check(*systems[k])
    F, S = systems[1]
    F[3, 3] += 0.25; F[7, 2] -= 0.1; F[2, 7] -= 0.1        # changed in place: a different system under the same array object
    check(F, S)
    check(*systems[2]); check(F, S)
    Ff = F.astype(np.complex128); Sf = S.astype(np.complex128)
    Ff.setflags(write=False); Sf.setflags(write=False)
    check(Ff, Sf); check(*systems[0]); check(Ff, Sf)
    assert engine._c128_keyed(Ff)[1] > 0 and engine._c128_keyed(Ff)[0] is Ff


def test_resident_systems_are_recognised_bitwise(engine):
    """negf_set_system keeps the last two systems on the device and re-selects one it recognises bit for bit: alternating
    between two systems (the spin blocks of a blockdiag Fock matrix), a third one evicting the least recently used, and a
    matrix changed IN PLACE by the caller must each give the result of their own (F, S) -- with one constant provider
    (whose F + Sigma_tot the library refreshes on every switch) serving all of them."""
    from gaunegf_amd.integrate import GrInt
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
    N = 40
    systems = [random_system(N, 60 + k) for k in range(3)]
    inds, s1, s2 = const_sigma_pair(N, systems[0][1], 4, 0.1)
    sc = SigmaCalculator(s1, s2)
    E = np.linspace(-1.0, 1.0, 9)
    w = np.full(9, 0.25) + 0.0j

    def ref_T(F, S):
        out = []
        for e in E:
            G = np.linalg.inv(e * S - F - s1 - s2)
            g1, g2 = 1j * (s1 - s1.conj().T), 1j * (s2 - s2.conj().T)
            out.append(np.real(np.trace(g1 @ G @ g2 @ G.conj().T)))
        return np.array(out)
    for k in (0, 1, 0, 1, 2, 0, 1, 2, 2):
        F, S = systems[k]
        assert np.allclose(calculate_transmission(F, S, sc, E), ref_T(F, S), rtol=1e-9, atol=1e-12), k
    F, S = systems[1]
    before = calculate_transmission(F, S, sc, E)
    F[3, 3] += 0.25                                     # the caller's array changes in place: a different system
    after = calculate_transmission(F, S, sc, E)
    assert np.allclose(after, ref_T(F, S), rtol=1e-9, atol=1e-12) and not np.allclose(after, before, rtol=1e-6)
    # and through a provider object that carries its own constant self-energies
    from gaunegf_amd.surfGTester import surfGTest
    for k in (0, 2, 0):
        F, S = systems[k]
        g_dev = surfGTest(F, S, inds, -0.1j)
        g_ref = oracle.ConstSigma(F, S, inds, -0.1j)
        assert rel_fro(GrInt(F, S, g_dev, E + 0.05j, w), oracle.GrInt(F, S, g_ref, E + 0.05j, w)) < TOL, k


@pytest.mark.parametrize("N", [2100, 4200])
def test_largest_window_configurations(engine, N):
    """n > 2048 runs the windowed path with 8 rows per lane and sub-panels of 4 columns, n > 4096 with
    16 rows per lane and sub-panels of 2; one energy, checked through the residual G A = I (an oracle
    inverse of this size costs seconds of CPU)."""
    from gaunegf_amd.integrate import GrBatch
    F, S, g_dev, g_ref = _const_provider(N, 77, nc=30)
    E = np.array([0.4 + 0.05j])
    G = GrBatch(F, S, g_dev, E)[0]
    A = E[0] * S - F - np.asarray(g_ref.sigmaTot(E[0]))
    R = G @ A - np.eye(N)
    assert np.linalg.norm(R) / np.sqrt(N) < 1e-9
    assert engine.last_info[0] == 0


def test_singular_matrix_is_reported(engine):
    from gaunegf_amd.integrate import GrInt
    from gaunegf_amd.surfGTester import surfGTest
    N = 6
    F = np.zeros((N, N)); S = np.eye(N)

    class Zero:
        def sigmaTot(self, E): return np.zeros((N, N), dtype=complex)
        def sigma(self, E, i): return np.zeros((N, N), dtype=complex)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        GrInt(F, S, Zero(), np.array([0.0, 1.0]), np.array([1.0, 1.0]))     # E=0: A == 0 exactly
    assert any("singular" in str(r.message) for r in rec)
    assert engine.last_info[0] == 1 and engine.last_info[1] == 0


@pytest.mark.parametrize("N", [64, 200, 300, 400])
def test_singular_and_nan_matrices_in_blocked_kernels(engine, N):
    """An exactly singular matrix (E S - F - Sigma == 0) and a NaN matrix in the middle of a batch:
    reported through info, NaN-filled, and the neighbouring energies are untouched."""
    from gaunegf_amd.integrate import GrBatch

    class Probe:                          # duck-typed provider (integrate.py:169): Sigma = 0 or NaN
        def __init__(self, bad): self.bad = bad
        def sigmaTot(self, E):
            z = np.zeros((N, N), dtype=complex)
            if self.bad == "nan" and abs(E - 1.0) < 1e-12:
                z[:, 3] = np.nan
            return z
        def sigma(self, E, i): return np.zeros((N, N), dtype=complex)

    S = np.eye(N)
    Fz, _ = random_system(N, 7)
    E = np.array([0.5 + 0.1j, 1.0 + 0j, 2.0 + 0.1j])
    for bad, F in (("singular", np.eye(N)), ("nan", Fz)):     # E=1: 1*I - I == 0
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            G = GrBatch(F, S, Probe(bad), E)
        assert np.all(np.isnan(G[1])), bad
        assert engine.last_info[1] != 0 and engine.last_info[0] == 0 and engine.last_info[2] == 0
        for k in (0, 2):
            ref = np.linalg.inv(E[k] * S - F)
            assert rel_fro(G[k], ref) < TOL, (bad, k)


def test_transmission_and_dos_golden(engine, golden_num):
    """tests/jax_optimization_suite.py: kernels with explicit sigma / gamma matrices."""
    from gaunegf_amd.transport import _transmission_kernel_restricted, _dos_kernel
    from gaunegf_amd.density import _compute_dos_at_energy
    g = golden_num
    F, S, st, G1, G2 = (g[f"rts24_{k}"] for k in ("F", "S", "sigma", "G1", "G2"))
    for E, T, D in zip(g["rts24_energies"], g["rts24_T"], g["rts24_dos"]):
        assert abs(_transmission_kernel_restricted(E, F, S, st, G1, G2) - T) < 1e-8 * max(1.0, abs(T))
        tot, site = _dos_kernel(E, F, S, st)
        assert abs(tot - D) < 1e-8 * max(1.0, abs(D))
        assert abs(np.sum(site) - tot) < 1e-9 * max(1.0, abs(tot))
        assert abs(_compute_dos_at_energy(E, F, S, st) - D) < 1e-8 * max(1.0, abs(D))


@pytest.mark.parametrize("N", [20, 64, 150])
def test_transmission_front_end(engine, N, tmp_path):
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission, calculate_dos, calculate_current
    F, S = random_system(N, 40 + N)
    inds, s1, s2 = const_sigma_pair(N, S, max(2, N // 10))
    sc = SigmaCalculator(s1, s2)
    E = np.linspace(-2, 2, 33)
    T = calculate_transmission(F, S, sc, E)
    g1 = sc.get_gamma(0, 0); g2 = sc.get_gamma(0, -1); st = sc.get_sigma_total(0)
    ref = np.array([oracle.transmission_restricted(e, F, S, st, g1, g2) for e in E])
    assert np.max(np.abs(T - ref) / np.maximum(1.0, np.abs(ref))) < TOL
    assert np.all(T >= -1e-12) and np.all(np.isfinite(T))              # test_transport_checkpointing.py:331
    # single-point == batch (rtol 1e-10, :343)
    from gaunegf_amd.transport import transmission_single_energy
    assert abs(transmission_single_energy(E[5], F, S, sc) - T[5]) <= 1e-10 * abs(T[5])
    # checkpoint resume equality (:396)
    ck = str(tmp_path / "ck.npz")
    T1 = calculate_transmission(F, S, sc, E, checkpoint_file=ck, checkpoint_interval=7)
    d = np.load(ck); part = d["transmission"].copy(); part[10:20] = -1
    np.savez(ck, transmission=part, energy_list=E)
    T2 = calculate_transmission(F, S, sc, E, checkpoint_file=ck, checkpoint_interval=7)
    assert np.allclose(T1, T2, rtol=1e-10, atol=0) and np.allclose(T1, T, rtol=1e-10, atol=0)
    # DOS: total = sum of sites (rtol 1e-9, :432), value vs oracle
    tot, site = calculate_dos(F, S, sc, E)
    assert np.allclose(site.sum(axis=1), tot, rtol=1e-9)
    dref = np.array([oracle.dos_kernel(e, F, S, st)[0] for e in E])
    assert np.max(np.abs(tot - dref) / np.maximum(1.0, np.abs(dref))) < TOL
    # current: zero bias -> 0 (:520), value vs oracle
    Ip = calculate_current(F, S, sc, 0.0, 0.1, T=0.0, dE=0.01)
    assert calculate_current(F, S, sc, 0.0, 0.0) == 0.0
    grid, muL, muR = oracle.current_grid(0.0, 0.1, 0.0, 0.01)
    Tg = np.array([oracle.transmission_restricted(e, F, S, st, g1, g2) for e in grid])
    assert abs(Ip - oracle.current_from_transmission(Tg, grid, muL, muR, 0.0, 'r')) < 1e-8 * abs(Ip)


def test_wire_current_symmetry(engine, golden_num):
    """The reference's tight-binding wire (tests/test_transport_checkpointing.py:22-58:
    t=-1, eps=0, S=I) with diagonal contacts is particle-hole symmetric, so
    |I(+V)| == |I(-V)| (rtol 1e-6, :549) and T >= 0 (:331)."""
    from gaunegf_amd.transport import SigmaCalculator, calculate_current, calculate_transmission
    F = np.real(golden_num["wire20_F"]); S = np.real(golden_num["wire20_S"])
    N = F.shape[0]
    inds, s1, s2 = const_sigma_pair(N, S, N // 2, gamma=0.1)       # contacts = N/2 sites each (:623)
    sc = SigmaCalculator(s1, s2)
    for T in (0.0, 300.0):
        Ip = calculate_current(F, S, sc, 0.0, 0.1, T=T, dE=0.005)
        Im = calculate_current(F, S, sc, 0.0, -0.1, T=T, dE=0.005)
        assert Ip > 0 and abs(abs(Ip) - abs(Im)) <= 1e-6 * abs(Ip)
    Tr = calculate_transmission(F, S, sc, np.linspace(-2.5, 2.5, 41))
    assert np.all(Tr >= 0) and np.all(np.isfinite(Tr))
    assert np.allclose(Tr, Tr[::-1], rtol=1e-9)


def test_legacy_wrappers(engine, tmp_path, capsys):
    """transport.py:724-1107 adapters: same numbers as the batch front-ends."""
    import scipy.io as io
    import gaunegf_amd.transport as T
    N = 16
    F, S = random_system(N, 5)
    inds, s1, s2 = const_sigma_pair(N, S, 3)
    E = np.linspace(-1, 1, 5)
    sc = T.SigmaCalculator(s1, s2)
    ref = T.calculate_transmission(F, S, sc, E)
    assert np.array_equal(np.array(T.cohTrans(E, F, S, s1, s2)), ref)
    tot, site = T.DOS(E, F, S, s1, s2)
    tr, sr = T.calculate_dos(F, S, sc, E)
    assert np.array_equal(np.array(tot), tr) and np.array_equal(site, sr)
    I = T.current(F, S, s1, s2, 0.0, 0.1, dE=0.01)
    assert I == T.calculate_current(F, S, sc, 0.0, 0.1, dE=0.01)
    assert T.currentSpin(F, S, s1, s2, 0.0, 0.1, dE=0.01) == [0, 0, 0, 0]
    fn = str(tmp_path / "scf.mat")
    io.savemat(fn, {"F": F, "S": S, "sig1": s1, "sig2": s2, "fermi": 0.0, "qV": 0.1, "spin": "r"})
    assert abs(T.currentF(fn, dE=0.01) - I) <= 1e-12 * abs(I)
    # energy-dependent adapters with a device-side provider
    from gaunegf_amd.surfGTester import surfGTest
    g = surfGTest(F, S, inds, -0.1j)
    assert np.allclose(T.cohTransE(E, F, S, g), ref, rtol=1e-12)
    assert np.allclose(T.DOSE(E, F, S, g)[0], tr, rtol=1e-12)
    assert abs(T.currentE(F, S, g, 0.0, 0.1, dE=0.01) - I) <= 1e-10 * abs(I)
    Fu = np.kron(np.eye(2), F); Su = np.kron(np.eye(2), S)
    Tt, Ts = T.cohTransSpin(E, Fu, Su, s1, s2, spin='u')
    assert np.allclose(Ts[:, 0], ref, rtol=1e-9) and np.allclose(Ts[:, 3], ref, rtol=1e-9)
    assert np.allclose(Ts[:, 1], 0, atol=1e-12) and np.allclose(Tt, 2 * ref, rtol=1e-9)


@pytest.mark.parametrize("spin", ["u", "g"])
def test_spin_block_transmission(engine, spin):
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
    N = 14
    rng = np.random.default_rng(9)
    Fa, Sa = random_system(N, 61)
    Fb, _ = random_system(N, 62)
    C = 0.05 * rng.standard_normal((N, N)); C = C + C.T
    # block form: [[alpha, C],[C, beta]] (a little spin mixing so that ud/du are non-zero)
    Fblk = np.block([[Fa, C], [C, Fb]]); Sblk = np.kron(np.eye(2), Sa)
    inds, s1, s2 = const_sigma_pair(N, Sa, 3)
    sc = SigmaCalculator(s1, s2)
    E = np.linspace(-1.5, 1.5, 7)
    if spin == 'u':
        F, S = Fblk, Sblk
    else:
        perm = np.concatenate([np.arange(0, 2 * N, 2), np.arange(1, 2 * N, 2)])
        inv = np.argsort(perm)
        F, S = Fblk[np.ix_(inv, inv)], Sblk[np.ix_(inv, inv)]        # spinor layout
    T, Ts = calculate_transmission(F, S, sc, E, spin=spin)
    st = np.kron(np.eye(2), s1 + s2)
    g1 = np.kron(np.eye(2), 1j * (s1 - s1.conj().T)); g2 = np.kron(np.eye(2), 1j * (s2 - s2.conj().T))
    for k, e in enumerate(E):
        tot, comp = oracle.transmission_spin_block(e, Fblk, Sblk, st, g1, g2)
        assert np.max(np.abs(Ts[k] - comp)) < TOL * max(1.0, np.max(np.abs(comp)))
        assert abs(T[k] - tot) < TOL * max(1.0, abs(tot))


# --------------------------------------------------------------------------- #
# 1-D chain decimation
# --------------------------------------------------------------------------- #
def _chain_system(N, nc, seed, eta):
    from gaunegf_amd.surfG1D import surfG
    F, S = random_system(N, seed)
    left = list(range(nc)); right = list(range(N - nc, N))
    aL = chain_lead(nc, seed + 1); aR = chain_lead(nc, seed + 2)
    taus = [aL[2].copy(), aR[2].copy()]; staus = [aL[3].copy(), aR[3].copy()]
    args = dict(taus=taus, staus=staus, alphas=[aL[0], aR[0]], aOverlaps=[aL[1], aR[1]],
                betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=eta)
    g_dev = surfG(F, S, [left, right], **args)
    g_ref = oracle.Chain1DSigma(F, S, [left, right], taus, staus, [aL[0], aR[0]], [aL[1], aR[1]],
                                [aL[2], aR[2]], [aL[3], aR[3]], eta=eta)
    return F, S, g_dev, g_ref


def test_compact_gamma_products_match_dense_and_oracle(engine):
    """G Gamma G^H and Tr[Gamma_L G Gamma_R G^H] through the contact columns / block of G only
    (negf_set_gamma_algo 0) against the dense n x n products (1) and the oracle, for a constant
    provider (support detected from the matrices) and a 1-D chain provider (contact blocks)."""
    from gaunegf_amd.integrate import GrLessInt
    from gaunegf_amd.transport import _transmission_kernel_restricted
    N, nc = 96, 12
    E = np.linspace(-2, 2, 24); w = np.full(24, 4.0 / 24) + 0j
    F, S, g_dev, g_ref = _const_provider(N, 321)
    Fc, Sc, gc_dev, gc_ref = _chain_system(N, nc, 77, 1e-3)
    gc_dev.force_iters = 25; gc_ref.force_iters = 25
    for (f, s, gd, gr) in ((F, S, g_dev, g_ref), (Fc, Sc, gc_dev, gc_ref)):
        for ind in (None, 0, -1):
            res = {}
            for algo in (0, 1):
                engine.set_gamma_algo(algo)
                try:
                    res[algo] = GrLessInt(f, s, gd, E, w, ind)
                finally:
                    engine.set_gamma_algo(0)
            ref = oracle.GrLessInt(f, s, gr, E, w, ind)
            assert rel_fro(res[0], res[1]) < 1e-12, ind
            assert rel_fro(res[0], ref) < TOL and rel_fro(res[1], ref) < TOL, ind
        h = gd._negf_lower(engine)
        engine.set_system(f, s)
        T = {}
        for algo in (0, 1):
            engine.set_gamma_algo(algo)
            try:
                T[algo] = engine.transmission(h, 0, 1, E)
            finally:
                engine.set_gamma_algo(0)
        assert np.max(np.abs(T[0] - T[1])) < 1e-12 * max(1.0, np.max(np.abs(T[1])))
        for k in (0, 11, 23):
            sg = [np.asarray(gr.sigma(E[k], i)) for i in (0, 1)]
            gam = [1j * (x - x.conj().T) for x in sg]
            tref = oracle.transmission_restricted(E[k], f, s, sg[0] + sg[1], gam[0], gam[1])
            assert abs(T[0][k] - tref) < 1e-8 * max(1.0, abs(tref)), k


@pytest.mark.parametrize("N", [200, 330, 650, 720])
def test_dense_hermitian_products_read_a_stored_conjugate_transpose(engine, N):
    """The dense products X G^H (G Gamma G^H, integrate.py:79-81; Tr[Gamma_L G Gamma_R G^H],
    transport.py:156-157) are taken as G X^H with X^H stored by the first product (both second operands plain) instead of
    conjugate-transposing tiles of G on the fly; the Hermitian form still computes the upper block tiles only and
    mirrors the rest (N = 200: the flexible-block kernel; 330: 6 x 6 block tiles, an edge block of 10 columns; 650: 11 x 11,
    the odd enumeration; 720: 12 x 12, edge 16).
    Dense products forced (negf_set_gamma_algo 1) against the compact path and the oracle."""
    from gaunegf_amd.integrate import GrLessInt
    E = np.linspace(-2, 2, 6); w = np.full(6, 4.0 / 6) + 0j
    F, S, g_dev, g_ref = _const_provider(N, 77 + N)
    for ind in (None, 0, -1):
        res = {}
        for algo in (0, 1):
            engine.set_gamma_algo(algo)
            try:
                res[algo] = GrLessInt(F, S, g_dev, E, w, ind)
            finally:
                engine.set_gamma_algo(0)
        ref = oracle.GrLessInt(F, S, g_ref, E, w, ind)
        assert rel_fro(res[0], res[1]) < 1e-12, ind
        assert rel_fro(res[1], ref) < TOL, ind
        assert rel_fro(res[1], res[1].conj().T) < 1e-13, ind          # (the mirrored half)
    h = g_dev._negf_lower(engine)
    engine.set_system(F, S)
    T = {}
    for algo in (0, 1):
        engine.set_gamma_algo(algo)
        try:
            T[algo] = engine.transmission(h, 0, 1, E)
        finally:
            engine.set_gamma_algo(0)
    assert np.max(np.abs(T[0] - T[1])) < 1e-12 * max(1.0, np.max(np.abs(T[1])))
    for k in (0, 5):
        sg = [np.asarray(g_ref.sigma(E[k], i)) for i in (0, 1)]
        gam = [1j * (x - x.conj().T) for x in sg]
        tref = oracle.transmission_restricted(E[k], F, S, sg[0] + sg[1], gam[0], gam[1])
        assert abs(T[1][k] - tref) < 1e-8 * max(1.0, abs(tref)), k


@pytest.mark.parametrize("nc", [4, 8, 9, 10, 16, 17, 18, 20, 24, 25, 33, 35, 40, 41, 48, 49, 50, 51, 56, 57, 63, 64, 65, 80])
def test_chain1d_fixed_trip_count(engine, nc):
    """Same number of sweeps on both sides -> the iterate itself must agree.  n_c = 33/48 and 50/64
    are the three- and four-tile kernels (50 = BASELINE C3's lead); 17-19, 33-35, 49-51 run their last tile as a
    remainder strip on the 4x4x4 matrix instruction; 65 and 80 the global-scratch kernel for n_c > 64.  The
    panels of the small inverse are 8 columns wide (half a column tile): 8, 16, 24, 40, 56 end on a panel boundary,
    9, 17, 25, 41, 49, 57 start a panel of one column, 63 ends one column short -- every pitch class and both
    halves of a tile as the last panel."""
    N = 3 * nc
    F, S, g_dev, g_ref = _chain_system(N, nc, 70 + nc, 1e-4)
    g_dev.force_iters = 40; g_ref.force_iters = 40
    for E in (0.3, -0.8, 0.1 + 0.2j):
        for i in (0, 1, -1):
            assert rel_fro(g_dev.sigma(E, i), g_ref.sigma(E, i)) < 1e-10, (nc, E, i)
        assert rel_fro(g_dev.sigmaTot(E), g_ref.sigmaTot(E)) < 1e-10
        gi, _, _ = oracle.chain1d_g(E, g_ref.aList[0], g_ref.aSList[0], g_ref.bList[0], g_ref.bSList[0],
                                    1e-4, force_iters=40)
        assert rel_fro(g_dev.g(E, 0), gi) < 1e-10


def test_chain1d_free_running(engine):
    """Reference stopping rule: iteration counts agree (+-1 at a threshold crossing) and
    Sigma within 10*conv relative."""
    nc = 8
    F, S, g_dev, g_ref = _chain_system(3 * nc, nc, 91, 1e-3)
    E = np.linspace(-1.5, 1.5, 16)
    sig, iters, conv = g_dev.sigma_batch(E)
    for k, e in enumerate(E):
        ref = g_ref.sigmaTot(e)
        c0 = g_ref.last_iters[(complex(e), 0)][0]; c1 = g_ref.last_iters[(complex(e), 1)][0]
        assert abs(int(iters[k, 0]) - c0) <= 1 and abs(int(iters[k, 1]) - c1) <= 1
        assert rel_fro(sig[k], ref) < 10 * 1e-5
    # and G(E) given the device Sigma: GrInt with the native provider vs oracle with
    # the oracle's own free-running Sigma (differences bounded by the fixed-point tolerance)
    from gaunegf_amd.integrate import GrInt
    w = np.ones_like(E) * (E[1] - E[0])
    assert rel_fro(GrInt(F, S, g_dev, E, w), oracle.GrInt(F, S, g_ref, E, w)) < 1e-3


def test_chain1d_order_predicted_for_new_grids(engine):
    """The chain kernel starts its jobs longest first; the lengths are predicted from the previous evaluation of the
    provider, for each energy the count of the nearest energy evaluated then -- also when the new grid has another size
    (an adaptive grid doubling, a Fermi search moving its contour).  The launch order must never change a result: every
    grid gives Sigma, the sweep counts and the convergence flags of a fresh provider's first (launch-order) evaluation,
    bit for bit."""
    nc = 10
    grids = [np.linspace(-1.5, 1.5, 37), np.linspace(-1.4, 1.6, 64), np.linspace(-1.5, 1.5, 37) + 0.01,
             np.linspace(-0.2, 0.2, 5), np.linspace(-1.5, 1.5, 37), np.linspace(-1.6, 1.4, 130) + 0.05j]
    _, _, g_seq, _ = _chain_system(3 * nc, nc, 77, 1e-3)
    engine.set_chain_cache(0)                                   # every evaluation runs the fixed point (no g(E) cache hits)
    engine.set_chain_round_robin(0, 0)                          # (the round-robin launch needs and takes no order)
    try:
        for E in grids:
            _, _, g_fresh, _ = _chain_system(3 * nc, nc, 77, 1e-3)
            sig0, it0, cv0 = g_fresh.sigma_batch(E)             # first evaluation of a provider: launch order
            sig1, it1, cv1 = g_seq.sigma_batch(E)               # order predicted from the grid before
            assert np.array_equal(it0, it1) and np.array_equal(cv0, cv1) and np.array_equal(sig0, sig1), E.size
            assert it1.max() > it1.min()                        # (the jobs do differ in length)
    finally:
        engine.set_chain_round_robin(-1, 0)
        engine.set_chain_cache(512)


def test_chain1d_order_prediction_over_batch_chunks(engine):
    """A grid evaluated in several batch chunks (negf_set_batch) records the WHOLE evaluation for the next prediction
    (chunks appended; the chunk at position 0 of the next evaluation promotes the record) -- and, as always, the order
    changes no result: chunked evaluations of the same and of a moved grid equal the unchunked first evaluation of a
    fresh provider bit for bit."""
    nc = 10
    E1 = np.linspace(-1.5, 1.5, 50)
    grids = [E1, E1, E1 + 0.013, np.linspace(-1.2, 1.7, 23), E1]
    _, _, g_seq, _ = _chain_system(3 * nc, nc, 78, 1e-3)
    engine.set_chain_cache(0)
    engine.set_chain_round_robin(0, 0)
    try:
        for E in grids:
            engine.set_batch(0)
            _, _, g_fresh, _ = _chain_system(3 * nc, nc, 78, 1e-3)
            sig0, it0, cv0 = g_fresh.sigma_batch(E)
            engine.set_batch(16)                                # 50 energies -> chunks of 16, 16, 16, 2
            sig1, it1, cv1 = g_seq.sigma_batch(E)
            assert np.array_equal(it0, it1) and np.array_equal(cv0, cv1) and np.array_equal(sig0, sig1), E.size
    finally:
        engine.set_batch(0)
        engine.set_chain_round_robin(-1, 0)
        engine.set_chain_cache(512)


@pytest.mark.parametrize("nc,eta", [(50, 1e-4), (64, 1e-3), (72, 1e-3)])
def test_chain1d_free_running_large_leads(engine, nc, eta):
    """The reference's stopping rule (surfG1D.py:271-288) at BASELINE C3's lead size and at the two
    kernel boundaries: iteration counts within +-1 (also at the 2000 cap), Sigma within 10*conv."""
    F, S, g_dev, g_ref = _chain_system(2 * nc + 20, nc, 5 + nc, eta)
    E = np.array([-1.7, -0.4, 0.9, 0.2 + 0.3j])
    sig, iters, conv = g_dev.sigma_batch(E)
    for k, e in enumerate(E):
        ref = g_ref.sigmaTot(e)
        for c in (0, 1):
            cnt, dlast = g_ref.last_iters[(complex(e), c)][:2]
            assert abs(int(iters[k, c]) - cnt) <= 1, (nc, e, c, int(iters[k, c]), cnt)
        assert rel_fro(sig[k], ref) < 10 * 1e-5, (nc, e)


def test_chain1d_forced_global_kernel(engine, monkeypatch):
    """NEGF_CHAIN1D_ALGO=global sends a small lead through the n_c > 64 kernel too: both kernels
    must give the same iterate as the oracle."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import oracle
        from test_gpu_parity import _chain_system
        from helpers import rel_fro
        F, S, g_dev, g_ref = _chain_system(60, 20, 90, 1e-4)
        g_dev.force_iters = 40; g_ref.force_iters = 40
        for E in (0.3, 0.1 + 0.2j):
            assert rel_fro(g_dev.sigmaTot(E), g_ref.sigmaTot(E)) < 1e-10
        print("ok")
    """) % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__))
    env = dict(os.environ, NEGF_CHAIN1D_ALGO="global")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_chain1d_integrals_fixed_trip(engine):
    """GrInt / GrLessInt / transmission with the device-side CHAIN1D provider at a fixed
    trip count: identical Sigma(E) inputs -> 1e-8 on the integrals."""
    from gaunegf_amd.integrate import GrInt, GrLessInt
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
    nc = 6
    F, S, g_dev, g_ref = _chain_system(30, nc, 33, 1e-4)
    g_dev.force_iters = 60; g_ref.force_iters = 60
    E, w = oracle.bias_window_grid(-0.3, 0.3, 24, 300.0)
    assert rel_fro(GrInt(F, S, g_dev, E, w), oracle.GrInt(F, S, g_ref, E, w)) < TOL
    for ind in (None, 0, -1):
        assert rel_fro(GrLessInt(F, S, g_dev, E, w, ind), oracle.GrLessInt(F, S, g_ref, E, w, ind)) < TOL
    T = calculate_transmission(F, S, SigmaCalculator(g_dev), E)
    for k, e in enumerate(E):
        s0 = g_ref.sigma(e, 0); s1 = g_ref.sigma(e, -1)
        ref = oracle.transmission_restricted(e, F, S, g_ref.sigmaTot(e), 1j * (s0 - s0.conj().T),
                                             1j * (s1 - s1.conj().T))
        assert abs(T[k] - ref) < TOL * max(1.0, abs(ref))


def test_surfG_from_fock_patterns_and_setF(engine):
    """Construction from F / S alone (surfG1D.py:131-147) and setF (:297-329) end to end: the device
    self-energy equals the oracle's for the blocks the constructor extracted, before and after a new F."""
    from gaunegf_amd.surfG1D import surfG
    N = 26
    F, S = random_system(N, 8)
    inds = [[0, 1, 2, 3], [22, 23, 24, 25]]
    conn = [[4, 5, 6, 7], [18, 19, 20, 21]]
    for taus in (None, conn):
        g = surfG(F, S, inds, taus=taus, eta=1e-3)
        g.force_iters = 20

        def ref_of(g):
            r = oracle.Chain1DSigma(g.F, g.S, inds, g.tauList, g.stauList, g.aList, g.aSList, g.bList, g.bSList,
                                    eta=1e-3)
            r.force_iters = 20
            return r
        for E in (0.2, -0.7 + 0.1j):
            assert rel_fro(g.sigmaTot(E), ref_of(g).sigmaTot(E)) < 1e-10, (taus, E)
        rng = np.random.default_rng(1)
        D = rng.standard_normal((N, N)); F2 = F + 0.05 * (D + D.T)
        g.setF(F2, 0.0, 0.0)
        for E in (0.2, -0.7 + 0.1j):
            assert rel_fro(g.sigmaTot(E), ref_of(g).sigmaTot(E)) < 1e-10, ("setF", taus, E)
            assert rel_fro(g.sigma(E, -1), ref_of(g).sigma(E, 1)) < 1e-10


def test_analytic_density_equals_grid_density(engine, golden_analytic):
    """The reference's closed-form density for constant contacts (density.py:276-329, vectors produced by
    executing it) against the GPU's real-axis + contour integrals of the same system: an end-to-end check of
    the contour orientation and prefactors that does not pass through the restated grid code's oracle."""
    from gaunegf_amd.density import densityRealN, densityComplexN
    g = golden_analytic
    F, S, s1, s2, X = g["an_F"], g["an_S"], g["an_sig1"], g["an_sig2"], g["an_X"]

    class Const:                                     # duck-typed constant provider (host-callback path)
        def sigmaTot(self, E): return s1 + s2
        def sigma(self, E, i): return (s1, s2)[i]
        def setF(self, *a): pass
    Eminf, Emin, mu = -1e6, -8.0, 0.2
    P = densityRealN(F, S, Const(), Eminf, Emin, 600, 0.0, showText=False) + \
        densityComplexN(F, S, Const(), Emin, mu, 486, 0.0, showText=False)
    Pan = np.real(X @ g["an0_Pbar"] @ X)
    assert rel_fro(P, Pan) < 2e-4
    assert abs(np.trace(P @ S) - np.trace(Pan @ S)) < 2e-4 * abs(np.trace(Pan @ S))


def test_perfect_wire_closed_form(engine):
    """Tight-binding chain t=-1, eps=0, S=I with exact 1-D leads: T(E)=1 inside the band
    (SURVEY.md section 8c closed form), to O(eta, conv)."""
    from gaunegf_amd.surfG1D import surfG
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
    N = 12
    F = np.zeros((N, N))
    for i in range(N - 1):
        F[i, i + 1] = F[i + 1, i] = -1.0
    S = np.eye(N)
    one = np.array([[0.0]]); t = np.array([[-1.0]]); z = np.array([[0.0]]); I1 = np.array([[1.0]])
    g = surfG(F, S, [[0], [N - 1]], taus=[t, t], staus=[z, z], alphas=[one, one], aOverlaps=[I1, I1],
              betas=[t, t], bOverlaps=[z, z], eta=1e-6)
    E = np.linspace(-1.5, 1.5, 11)
    T = calculate_transmission(F, S, SigmaCalculator(g), E)
    assert np.max(np.abs(T - 1.0)) < 5e-3
    # closed-form surface Green's function g_s = (E - i sqrt(4 - E^2)) / 2 for |E| < 2
    gs = g.g(0.5, 0, conv=1e-9)
    assert abs(gs[0, 0] - (0.5 - 1j * np.sqrt(4 - 0.25)) / 2) < 1e-3


# --------------------------------------------------------------------------- #
# Bethe lattice
# --------------------------------------------------------------------------- #
def _bethe_atom(name="Au"):
    from gaunegf_amd.surfGBethe import read_bethe_params, construct_sk_matrix, gen_neighbors, surfGBAt
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaunegf_amd", "data", name)
    ne, Ed, Vd, Sd, H0 = read_bethe_params(here)
    dirs = gen_neighbors(np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.2, 0.0]))
    Sl = [construct_sk_matrix(Sd, d) for d in dirs]
    Vl = [construct_sk_matrix(Vd, d) for d in dirs]
    return surfGBAt(H0, Sl, Vl, 1e-6), H0, Sl, Vl


@pytest.mark.parametrize("name", ["Au", "Au2"])
def test_bethe_raw_fixed_trip_count(engine, name):
    at, H0, Sl, Vl = _bethe_atom(name)
    at.force_iters = 25
    for E in (-5.0, 0.7, -2.0 + 0.3j):
        ref, _, _ = oracle.bethe_sigmaK(E, H0, Sl, Vl, 1e-6, force_iters=25)
        assert rel_fro(at.sigmaK(E), ref) < 1e-10, (name, E)
        ref9, _, _, _ = oracle.bethe_sigma_surface(E, H0, Sl, Vl, 1e-6, force_iters=25)
        assert rel_fro(at.sigma(E), ref9) < 1e-10, (name, E)
    cl = oracle.bethe_cluster_sigma_total(-5.0, H0, Sl, Vl, 1e-6, force_iters=25)
    assert rel_fro(at.sigmaTot(-5.0), cl) < 1e-10


@pytest.mark.parametrize("name", ["Au", "Au2"])
def test_bethe_surface_loop_given_device_bulk(engine, name):
    """The oracle's surface loop is pinned to the reference's numpy twin (surfG3D.py:907-979,
    test_oracle_golden.py); started from the DEVICE's own bulk self-energies it must reproduce the device's
    surface self-energies, and the pinned cluster assembly the device's cluster matrix."""
    at, H0, Sl, Vl = _bethe_atom(name)
    at.force_iters = 25
    for E in (-5.0, 0.7, -2.0 + 0.3j):
        sigK = at.sigmaK(E)
        ref9, _, _, _ = oracle.bethe_sigma_surface(E, H0, Sl, Vl, 1e-6, force_iters=25, sigK=sigK)
        assert rel_fro(at.sigma(E), ref9) < 1e-10, (name, E)
        assert rel_fro(at.sigmaTot(E), oracle.bethe_cluster_sigma_total(E, H0, Sl, Vl, 1e-6, sigK=sigK)) < 1e-13


def test_bethe_raw_free_running(engine):
    at, H0, Sl, Vl = _bethe_atom()
    E = np.array([-6.0, -5.0, -3.0, 1.0])
    out = at.sigmaK(E)
    its = at.last_iters.copy()
    for k, e in enumerate(E):
        ref, count, diff = oracle.bethe_sigmaK(e, H0, Sl, Vl, 1e-6)
        assert abs(int(its[k]) - count) <= 1
        assert rel_fro(out[k], ref) < 10 * 1e-5


def test_bethe_contact_fermi_level(engine, monkeypatch, capsys):
    """surfGBAt.calcFermi (surfGBethe.py:1158-1188): contour + real-axis integrals of the 13-site cluster with
    the Bethe self-energy.  The electron count on the centre site at the returned level -- recomputed with the
    ORACLE serving the same grids (oracle.GrInt, oracle Bethe loops) -- must equal ne within the search tolerance."""
    import gaunegf_amd.density as D
    from gaunegf_amd.config import ENERGY_MIN
    at, H0, Sl, Vl = _bethe_atom()
    at.eta = 1e-4
    ne, tol = 5.5, 5e-2                                         # Au: 11 electrons / 2 (surfGBethe.py:208)
    orbs = D._orbital_energies(at.F, at.S)
    f0 = (orbs[int(ne) - 1] + orbs[int(ne)]) / 2
    Emin, N1, N2 = D.integralFit(at.F, at.S, at, f0, ENERGY_MIN, tol, at.T, maxN=1000)
    Ef = D.calcFermi(at, ne, Emin, max(orbs), f0, N1, N2, ENERGY_MIN, at.T, tol, 1000, nOrbs=9)[0]
    assert Ef == at.calcFermi(ne, tol=tol)                     # the method is exactly this sequence
    assert Emin < Ef < max(orbs)

    class RefCluster:                                           # the same cluster served by the numpy oracle
        F, S = at.F, at.S
        def sigmaTot(self, E, conv=None):
            return oracle.bethe_cluster_sigma_total(E, H0, Sl, Vl, at.eta)
    ref = RefCluster()
    monkeypatch.setattr(D, "GrInt", oracle.GrInt)
    P = np.real(D.densityRealN(at.F, at.S, ref, ENERGY_MIN, Emin, int(N2), 0, showText=False) +
                D.densityComplexN(at.F, at.S, ref, Emin, Ef, int(N1), at.T, showText=False, method='legendre'))
    monkeypatch.undo()
    n_center = np.real(D._count(P, at.S, 9))
    assert abs(n_center - ne) < tol + 5e-3, (n_center, Ef)
    # cross-check the vectorised cluster self-energy against the per-energy protocol call
    Es = np.array([-3.0, 0.5])
    both = at.sigmaTot_batch(Es)
    for k, e in enumerate(Es):
        assert rel_fro(both[k], at.sigmaTot(e)) < 1e-12


def _bethe_device(name, N=40):
    from gaunegf_amd.surfGBethe import surfGB
    # two contacts of 2 atoms x 9 orbitals at the ends of an N-orbital device; geometry:
    # atoms on a line plus one in-plane neighbour each so that the surface normal is defined
    coords = np.array([[0, 0, 0.0], [2.88, 0, 0], [1.44, 2.494, 0],
                       [0, 0, 20.0], [2.88, 0, 20.0], [1.44, 2.494, 20.0],
                       [1.44, 0.8, 10.0]])
    n_atoms = len(coords)
    orbMap = np.concatenate([np.full(9, a + 1) for a in range(6)] + [np.full(N - 54, 7)])
    typ_one = np.array([0, 1001, 1002, 1003, 2001, 2002, 2003, 2004, 2005])
    orbTyp = np.concatenate([typ_one] * 6 + [np.zeros(N - 54, dtype=int)])
    return coords, orbMap, orbTyp


@pytest.mark.parametrize("name", ["Au", "Au2"])
def test_bethe_contact_assembly_and_integrals(engine, name):
    from gaunegf_amd.surfGBethe import surfGB
    from gaunegf_amd.integrate import GrInt, GrLessInt
    N = 60
    coords, orbMap, orbTyp = _bethe_device(name, N)
    F, S = random_system(N, 77)
    lat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaunegf_amd", "data", name)
    g = surfGB.from_arrays(F, S, [[1, 2, 3], [4, 5, 6]], orbMap, orbTyp, coords, latFile=lat, eta=1e-6, fermi=0.0)
    g.force_iters = 30
    Xi = g.Xi if g.Sdict['sss'] == 0 else None

    class Ref:
        def sigma(self, E, i, conv=None):
            at = g.gList[i]
            return oracle.bethe_contact_sigma(E, N, g.indsLists[i], g.nIndLists[i], at.H, at.Slist, at.Vlist,
                                              1e-6, Xi=Xi, force_iters=30)
        def sigmaTot(self, E, conv=None):
            return self.sigma(E, 0) + self.sigma(E, 1)
    ref = Ref()
    for E in (-4.0, 0.5):
        for i in (0, 1):
            assert rel_fro(g.sigma(E, i), ref.sigma(E, i)) < 1e-9, (name, E, i)
        assert rel_fro(g.sigmaTot(E), ref.sigmaTot(E)) < 1e-9
    E, w = oracle.contour_grid(-8.0, 0.0, 18, 0.0)
    assert rel_fro(GrInt(F, S, g, E, w), oracle.GrInt(F, S, ref, E, w)) < TOL
    Eg, wg = oracle.bias_window_grid(-0.2, 0.2, 8, 300.0)
    for ind in (None, 0, -1):
        assert rel_fro(GrLessInt(F, S, g, Eg, wg, ind), oracle.GrLessInt(F, S, ref, Eg, wg, ind)) < TOL


@pytest.mark.parametrize("spin", ["r", "u"])
def test_bethe_contact_assembly_given_device_surface(engine, spin):
    """SURVEY a16.  The oracle's contact assembly is pinned to the reference's numpy twin (surfG3D.py:417-433,
    test_oracle_golden.py::test_bethe_contact_assembly_vs_reference_numpy_twin); handed the DEVICE's own surface
    self-energies it must reproduce the device's assembled contact matrices -- also for attached directions
    outside 0..8, where both follow the jax indexing rules (wrap, then clamp)."""
    from gaunegf_amd.surfGBethe import surfGB
    N = 60
    coords, orbMap, orbTyp = _bethe_device("Au", N)
    F, S = random_system(N, 78)
    lat = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gaunegf_amd", "data", "Au")
    g = surfGB.from_arrays(F, S, [[1, 2, 3], [4, 5, 6]], orbMap, orbTyp, coords, latFile=lat, eta=1e-6, fermi=0.0, spin=spin)
    g.force_iters = 20
    for at in g.gList:
        at.force_iters = 20                              # the per-atom objects run the same fixed trip count
    assert g.Sdict['sss'] != 0
    for variant in range(2):
        if variant == 1:                                 # out-of-range and negative attached directions
            g.nIndLists = [[[0, 9], [-1, 3, 11]] + [list(v) for v in g.nIndLists[0][2:]], [list(v) for v in g.nIndLists[1]]]
            g._version += 1                              # the lowered provider is rebuilt
        for E in (-4.0, 0.5):
            tot = 0
            for i in (0, 1):
                s9 = g.gList[i].sigma(E)
                ref = oracle.bethe_contact_sigma(E, N, g.indsLists[i], g.nIndLists[i], None, None, None, 1e-6, spin=spin, sigSurf=s9)
                got = g.sigma(E, i)
                assert got.shape == ref.shape and rel_fro(got, ref) < 1e-13, (variant, E, i)
                tot = tot + ref
            assert rel_fro(g.sigmaTot(E), tot) < 1e-13


# --------------------------------------------------------------------------- #
# density front-ends end to end, and full-size properties
# --------------------------------------------------------------------------- #
def test_density_front_ends(engine):
    from gaunegf_amd.density import densityComplexN, densityRealN, densityGridN, densityComplex
    F, S, g_dev, g_ref = _const_provider(40, 5)
    P = densityComplexN(F, S, g_dev, -6.0, 0.1, 54, 300.0, showText=False)
    E, w = oracle.contour_grid(-6.0, 0.1, 54, 300.0); Eb, wb = oracle.broadening_grid(0.1, 54, 300.0)
    ref = np.imag(oracle.GrInt(F, S, g_ref, E, w) + oracle.GrInt(F, S, g_ref, Eb, wb)) / np.pi
    assert rel_fro(P, ref) < TOL
    Pr = densityRealN(F, S, g_dev, -50.0, -6.0, 16, 0.0, showText=False)
    E, w = oracle.real_axis_grid(-50.0, -6.0, 16, 0.0)
    assert rel_fro(Pr, -np.imag(oracle.GrInt(F, S, g_ref, E, w)) / np.pi) < TOL
    Pg = densityGridN(F, S, g_dev, -0.2, 0.3, ind=-1, N=20, T=300.0, showText=False)
    E, w = oracle.bias_window_grid(-0.2, 0.3, 20, 300.0)
    assert rel_fro(Pg, oracle.GrLessInt(F, S, g_ref, E, w, -1) / (2 * np.pi)) < TOL
    # adaptive contour: same level sequence as the oracle's driver -> same value
    Pa = densityComplex(F, S, g_dev, -6.0, 0.1, tol=1e-5, T=0.0)
    _, center, r = -6.0, (-6.0 + 0.1) / 2, (0.1 + 6.0) / 2

    def cp(x, wq):
        th = np.pi / 2 * (x + 1); z = center + r * np.exp(1j * th); dz = 1j * r * np.exp(1j * th)
        return oracle.GrInt(F, S, g_ref, z, (np.pi / 2) * wq * dz * oracle.fermi(z, 0.1, 0.0))
    assert rel_fro(Pa, np.imag(oracle.adaptive_ant(cp, tol=1e-5)) / np.pi) < 1e-7
    # trace of P S counts electrons: positive and below N
    ne = np.real(np.trace(P @ S))
    assert 0 < ne < 40


def test_saveMAT_round_trip_through_currentF(engine, tmp_path, capsys):
    """NEGFE.saveMAT (scf.py:823-843) writes the keys F, sig1, sig2, S, fermi, qV, spin, den, conv;
    transport.currentF (transport.py:847-875) reads that file back: the current from the file equals the current
    computed from the in-memory matrices, and the oracle's value."""
    import scipy.io as io
    from gaunegf_amd.scfE import NEGFE
    from gaunegf_amd.transport import currentF, current
    N = 24
    F, S, g_dev, g_ref = _const_provider(N, 12, nc=4)
    sys_ = NEGFE(F, S, g_dev, ne=2 * 9, spin='r', T=300.0, Eminf=-60.0)
    sys_.setIntegralLimits(N1=24, N2=8, Nnegf=8, tol=1e-4, Emin=-7.0)
    sys_.setVoltage(0.2, fermi=0.1)
    sys_.FockToP()
    fn = str(tmp_path / "out.mat")
    Fbar = sys_.saveMAT(fn)
    assert rel_fro(Fbar, sys_.X @ F @ sys_.X) < 1e-14
    m = io.loadmat(fn)
    for k in ("F", "sig1", "sig2", "S", "fermi", "qV", "spin", "den", "conv"):
        assert k in m, k
    assert np.array_equal(m["F"], F) and np.array_equal(m["S"], S) and m["spin"][0] == 'r'
    assert m["fermi"][0, 0] == 0.1 and m["qV"][0, 0] == 0.2 and rel_fro(m["den"], sys_.P) == 0.0
    s1, s2 = sys_.getSigma(0.1)
    assert np.array_equal(m["sig1"], s1) and np.array_equal(m["sig2"], s2)
    I_file = currentF(fn, dE=0.01, T=300.0)
    I_mem = current(F, S, s1, s2, 0.1, 0.2, 300.0, 'r', dE=0.01)
    assert I_file == I_mem and I_file != 0.0
    # oracle: the same grid and quadrature with the numpy transmission
    from gaunegf_amd.transport import calculate_current, SigmaCalculator
    import gaunegf_amd.transport as TR
    Eg = []
    orig = TR.calculate_transmission
    def spy(F_, S_, sc_, energies, spin=None, **kw):
        Eg.append(np.asarray(energies)); return orig(F_, S_, sc_, energies, spin=spin, **kw)
    TR.calculate_transmission = spy
    try:
        calculate_current(F, S, SigmaCalculator(s1, s2), 0.1, 0.2, 300.0, 'r', 0.01)
    finally:
        TR.calculate_transmission = orig
    sc = SigmaCalculator(s1, s2)
    Tref = np.array([oracle.transmission_restricted(e, F, S, sc.get_sigma_total(e), sc.get_gamma(e, 0), sc.get_gamma(e, -1))
                     for e in Eg[0]])
    Tdev = orig(F, S, sc, Eg[0], spin='r')
    assert np.max(np.abs(Tdev - Tref)) < 1e-9 * max(1.0, np.max(np.abs(Tref)))


def test_fock_to_p_density_step(engine, capsys):
    """scfE.NEGFE.FockToP (scfE.py:301-462) without Gaussian: with a given Fermi level the density
    matrix is the sum of the real-axis, contour (+ broadening) and bias-window integrals with the
    reference's prefactors -- checked against the same sum built from the oracle's grids and
    integrals; with the Fermi search switched on the electron count lands on the target."""
    from gaunegf_amd.scfE import NEGFE
    N = 24
    F, S, g_dev, g_ref = _const_provider(N, 11, nc=4)
    T, N1, N2, Nn = 300.0, 30, 12, 10
    Emin, Eminf, mu = -7.0, -60.0, 0.15

    def ref_P(mu1, mu2):
        E, w = oracle.real_axis_grid(Eminf, Emin, N2, 0.0)
        P = -np.imag(oracle.GrInt(F, S, g_ref, E, w)) / np.pi
        E, w = oracle.contour_grid(Emin, mu1, N1, T); Eb, wb = oracle.broadening_grid(mu1, N1, T)
        P = P + np.imag(oracle.GrInt(F, S, g_ref, E, w) + oracle.GrInt(F, S, g_ref, Eb, wb)) / np.pi
        if mu1 != mu2:
            E, w = oracle.bias_window_grid(mu1, mu2, Nn, T)
            P = P + oracle.GrLessInt(F, S, g_ref, E, w, -1) / (2 * np.pi)
        return P

    for qV in (0.0, 0.3):
        sys_ = NEGFE(F, S, g_dev, ne=2 * 9, spin='r', T=T, Eminf=Eminf)
        sys_.setIntegralLimits(N1=N1, N2=N2, Nnegf=Nn, tol=1e-4, Emin=Emin)
        sys_.setVoltage(qV, fermi=mu)
        sys_.Nnegf = Nn                                  # setVoltage resets the bias grid to its default of 50
        EList, occ = sys_.FockToP()
        assert rel_fro(sys_.P, ref_P(mu + qV / 2, mu - qV / 2)) < TOL, qV
        assert np.all(np.diff(EList) >= 0) and len(occ) == N
        assert abs(np.sum(occ) - np.real(np.trace(S @ sys_.P))) < 1e-9      # Lowdin occupations sum to Tr(S P)

    # Fermi search on: the returned level reproduces the electron target (restricted: ne/2 per spin)
    sys_ = NEGFE(F, S, g_dev, ne=2 * 9, spin='r', T=0.0, Eminf=Eminf)
    sys_.setIntegralLimits(N1=64, N2=N2, tol=1e-4, Emin=Emin)
    sys_.setVoltage(0.0, fermiMethod='muller')
    assert sys_.updFermi
    sys_.FockToP()
    assert abs(np.real(np.trace(S @ sys_.P)) - 9.0) < 5e-3
    # 'predict' (scfE.py:333-361): the constant-self-energy estimate moves the level towards the target
    sys_ = NEGFE(F, S, g_dev, ne=2 * 9, spin='r', T=0.0, Eminf=Eminf)
    sys_.setIntegralLimits(N1=64, N2=N2, tol=1e-4, Emin=Emin)
    sys_.setVoltage(0.0, fermiMethod='predict')
    start = sys_.fermi
    errs = []
    for _ in range(6):                                   # each step integrates up to the level the previous one set
        sys_.FockToP()
        errs.append(abs(np.real(np.trace(S @ sys_.P)) - 9.0))
    assert np.isfinite(sys_.fermi) and sys_.fermi != start
    # the predictor keeps correcting by the electrons the last density was off by: the count closes in
    assert errs[-1] < 0.1 and errs[-1] <= max(errs[:2]), errs

    # the SCF cycle (scf.py:663-800) on a model Fock builder F0 + U diag(P) (a Hubbard-like mean field):
    # damping alone, and damping with a Pulay/DIIS step every (nPulay + 1)-th cycle
    F0 = F.copy()
    hist = {}
    for pulay in (False, True):
        sys_ = NEGFE(F0, S, g_dev, ne=18, spin='r', T=T, Eminf=Eminf,
                     fock_builder=lambda P: F0 + 0.3 * np.diag(np.real(np.diag(P))))
        sys_.setIntegralLimits(N1=N1, N2=N2, tol=1e-4, Emin=Emin)
        sys_.setVoltage(0.0, fermi=mu)
        hist[pulay] = sys_.SCF(conv=1e-6, damping=0.5, maxcycles=60, pulay=pulay)
        assert hist[pulay][-1] < 1e-6, pulay
        P_fix = sys_.P
        sys_.FockToP()                                   # self-consistency: the density of the final Fock matrix
        assert np.max(np.abs(np.diag(sys_.P) - np.diag(P_fix))) < 1e-5
    assert len(hist[True]) <= len(hist[False])           # DIIS does not need more cycles than damping


def test_full_size_properties_C2(engine):
    """BASELINE config C2 (N=200, constant Sigma, 1000 energies): size-independent
    properties instead of a 1000-point oracle run."""
    from gaunegf_amd.integrate import GrInt, GrLessInt, GrBatch
    N, M = 200, 1000
    F, S, g_dev, g_ref = _const_provider(N, 2, nc=20)
    E, w = oracle.real_axis_grid(-3.0, 3.0, M, 0.0)
    rng = np.random.default_rng(0)
    w1 = rng.standard_normal(M); w2 = rng.standard_normal(M)
    a = GrInt(F, S, g_dev, E, w1); b = GrInt(F, S, g_dev, E, w2); c = GrInt(F, S, g_dev, E, w1 + 2 * w2)
    assert rel_fro(c, a + 2 * b) < 1e-12                       # linearity in the weights
    assert np.array_equal(a, GrInt(F, S, g_dev, E, w1))        # run-to-run bitwise reproducible
    # split grid = whole grid (batching / accumulation order independence to rounding)
    d = GrInt(F, S, g_dev, E[:333], w1[:333]) + GrInt(F, S, g_dev, E[333:], w1[333:])
    assert rel_fro(d, a) < 1e-12
    # residual of a sample of inverses: G (E S - F - Sigma) = I
    idx = rng.choice(M, 12, replace=False)
    G = GrBatch(F, S, g_dev, E[idx])
    st = g_ref.sigmaTot(0.0)
    for k, i in enumerate(idx):
        A = E[i] * S - F - st
        assert np.linalg.norm(G[k] @ A - np.eye(N)) / np.sqrt(N) < 1e-9
        assert rel_fro(G[k], oracle.gr_point(st, E[i], F, S)) < TOL
    # G Gamma G^H is Hermitian for a Hermitian Gamma and real weights
    L = GrLessInt(F, S, g_dev, E[:200], np.abs(w1[:200]), -1)
    assert rel_fro(L, L.conj().T) < 1e-12
    # bounded oracle sample of the integral itself
    sub = slice(0, M, 25)
    assert rel_fro(GrInt(F, S, g_dev, E[sub], w[sub]), oracle.GrInt(F, S, g_ref, E[sub], w[sub])) < TOL


# --------------------------------------------------------------------------- #
# BASELINE.json configs C3 / C4 / C5 at their real matrix sizes (bounded energy samples:
# the oracle costs seconds per point at these sizes)
# --------------------------------------------------------------------------- #
def test_config_C3_shape(engine):
    """C3: N=500 device, 1-D chain leads n_c=50, eta=1e-4 (examples/SiNEGF.py:44), a 16-point sample of
    the 2000-point Legendre grid on [-2,2] eV; fixed trip count -> identical Sigma(E) -> 1e-8 on GrInt,
    GrLessInt and the per-energy G."""
    from gaunegf_amd.integrate import GrInt, GrLessInt, GrBatch
    F, S, g_dev, g_ref = _chain_system(500, 50, 3, 1e-4)
    g_dev.force_iters = 50; g_ref.force_iters = 50
    E, w = oracle.real_axis_grid(-2.0, 2.0, 2000, 0.0)
    sub = np.arange(7, 2000, 125)
    assert len(sub) == 16
    ref = oracle.GrInt(F, S, g_ref, E[sub], w[sub])
    assert rel_fro(GrInt(F, S, g_dev, E[sub], w[sub]), ref) < TOL
    assert rel_fro(GrLessInt(F, S, g_dev, E[sub[:4]], w[sub[:4]], -1),
                   oracle.GrLessInt(F, S, g_ref, E[sub[:4]], w[sub[:4]], -1)) < TOL
    G = GrBatch(F, S, g_dev, E[sub[:2]])
    for k in range(2):
        assert rel_fro(G[k], oracle.gr_point(g_ref.sigmaTot(E[sub[k]]), E[sub[k]], F, S)) < TOL


def test_config_C3_free_running_at_the_sweep_cap(engine):
    """C3 at full size with the PRODUCTION stopping rule (surfG1D.py:271-288): 69 % of the headline grid's fixed
    points stop at the 2000-sweep cap, where the iterate is not a fixed point.  Six such energies and two that
    converge, picked from a 32-point probe of the 2000-point grid, against the oracle: sweep counts (+-1 only where
    the threshold is crossed; at the cap both run exactly 2000), Sigma(E), and GrInt on that sub-grid.  The grid is
    evaluated twice with the round robin off -- the second launch then runs longest-first in the order learned from
    the first -- and must reproduce Sigma and the counts bit for bit; then with the round robin forced on 5 slots."""
    from gaunegf_amd.integrate import GrInt
    F, S, g_dev, g_ref = _chain_system(500, 50, 3, 1e-4)
    E, w = oracle.real_axis_grid(-2.0, 2.0, 2000, 0.0)
    probe = np.arange(3, 2000, 63)
    _, it, _ = g_dev.sigma_batch(E[probe])
    at_cap = it.max(axis=1) >= 2000
    assert at_cap.sum() >= 6 and (~at_cap).sum() >= 2, it.max(axis=1)
    sub = np.concatenate([probe[at_cap][:6], probe[~at_cap][:2]])
    engine.set_chain_cache(0)                                     # all evaluations run the fixed point
    engine.set_chain_round_robin(0, 0)
    try:
        sig, iters, cv = g_dev.sigma_batch(E[sub])
        sig2, iters2, cv2 = g_dev.sigma_batch(E[sub])             # learned (longest-first) launch order
        engine.set_chain_round_robin(150, 5)
        sig5, iters5, cv5 = g_dev.sigma_batch(E[sub])             # 16 fixed points through 5 slots, quanta of 150 sweeps
    finally:
        engine.set_chain_round_robin(-1, 0)
        engine.set_chain_cache(512)
    assert np.array_equal(iters, iters2) and np.array_equal(cv, cv2) and np.array_equal(sig, sig2)
    assert np.array_equal(iters, iters5) and np.array_equal(cv, cv5) and np.array_equal(sig, sig5)
    sig3, iters3, cv3 = g_dev.sigma_batch(E[sub])                 # fills the g(E) cache ...
    sig4, iters4, cv4 = g_dev.sigma_batch(E[sub])                 # ... and is served from it: the same bits again
    assert engine.chain_cache_stats()["hits"] >= 1
    for s_, i_, c_ in ((sig3, iters3, cv3), (sig4, iters4, cv4)):
        assert np.array_equal(iters, i_) and np.array_equal(cv, c_) and np.array_equal(sig, s_)
    for k, e in enumerate(E[sub]):
        ref = g_ref.sigmaTot(e)
        for c in (0, 1):
            cnt = g_ref.last_iters[(complex(e), c)][0]
            assert abs(int(iters[k, c]) - cnt) <= (0 if cnt >= 2000 else 1), (e, c, int(iters[k, c]), cnt)
        capped = max(g_ref.last_iters[(complex(e), c)][0] for c in (0, 1)) >= 2000
        assert rel_fro(sig[k], ref) < (1e-7 if capped and iters[k].min() >= 2000 else 10 * 1e-5), (e, iters[k])
    assert rel_fro(GrInt(F, S, g_dev, E[sub], w[sub]), oracle.GrInt(F, S, g_ref, E[sub], w[sub])) < 1e-4


def test_config_C4_shape(engine):
    """C4: N=800, Bethe-lattice Sigma (Au.bethe), 2 contacts x 3 atoms x 9 orbitals; 3 points of the
    486-point ANT contour and 2 of the 256-point real-axis grid, fixed trip count."""
    from gaunegf_amd.surfGBethe import surfGB
    from gaunegf_amd.integrate import GrInt
    N = 800
    F, S = random_system(N, 4)
    coords, orbMap, orbTyp = _bethe_device("Au", N)
    lat = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gaunegf_amd", "data", "Au")
    g = surfGB.from_arrays(F, S, [[1, 2, 3], [4, 5, 6]], orbMap, orbTyp, coords, latFile=lat, eta=1e-6, fermi=0.0)
    g.force_iters = 30
    Xi = g.Xi if g.Sdict['sss'] == 0 else None

    class Ref:
        def sigma(self, E, i, conv=None):
            at = g.gList[i]
            return oracle.bethe_contact_sigma(E, N, g.indsLists[i], g.nIndLists[i], at.H, at.Slist, at.Vlist,
                                              1e-6, Xi=Xi, force_iters=30)
        def sigmaTot(self, E, conv=None): return self.sigma(E, 0) + self.sigma(E, 1)
    E, w = oracle.contour_grid(-8.0, 0.0, 486, 0.0)
    sub = np.array([0, 243, 485])
    assert rel_fro(GrInt(F, S, g, E[sub], w[sub]), oracle.GrInt(F, S, Ref(), E[sub], w[sub])) < TOL
    Er, wr = oracle.real_axis_grid(-1e6, -8.0, 256, 0.0)
    sub = np.array([3, 250])
    assert rel_fro(GrInt(F, S, g, Er[sub], wr[sub]), oracle.GrInt(F, S, Ref(), Er[sub], wr[sub])) < TOL


def test_spin_block_diagonal_fast_path(engine):
    """blockdiag(alpha, beta) systems (scf.py:177-180) run as two N-sized solves: GrInt / GrLessInt /
    spin-'u' transmission must equal the full 2N x 2N path to 1e-12 and the oracle to 1e-8; a system WITH
    spin mixing must not take the fast path."""
    import gaunegf_amd.integrate as I
    import gaunegf_amd.transport as T
    from gaunegf_amd.integrate import GrInt, GrLessInt
    from gaunegf_amd.surfGTester import surfGTest
    N, nc = 40, 5
    Fa, Sa = random_system(N, 71); Fb, _ = random_system(N, 72)
    Z = np.zeros((N, N))
    F = np.block([[Fa, Z], [Z, Fb]]); S = np.kron(np.eye(2), Sa)
    inds, s1, s2 = const_sigma_pair(N, Sa, nc)
    sig = [np.kron(np.eye(2), s1), np.kron(np.eye(2), s2)]

    class G:                                             # foreign provider on the 2N space (host callback)
        def sigma(self, E, i): return sig[i]
        def sigmaTot(self, E): return sig[0] + sig[1]
    inds2 = [list(inds[0]) + [N + i for i in inds[0]], list(inds[1]) + [N + i for i in inds[1]]]
    g_const = surfGTest(F, S, inds2, -0.1j)              # device-side constant provider on the 2N space
    E, w = oracle.bias_window_grid(-0.25, 0.25, 12, 300.0)
    res = {}
    for flag in (True, False):
        I.SPIN_BLOCK_SPLIT = T.SPIN_BLOCK_SPLIT = flag
        try:
            res[flag] = (GrInt(F, S, G(), E, w), GrLessInt(F, S, G(), E, w, -1), GrInt(F, S, g_const, E, w),
                         GrLessInt(F, S, g_const, E, w, 0),
                         T.calculate_transmission(F, S, T.SigmaCalculator(s1, s2), np.real(E), spin='u'))
        finally:
            I.SPIN_BLOCK_SPLIT = T.SPIN_BLOCK_SPLIT = True
    for k in range(4):
        assert rel_fro(res[True][k], res[False][k]) < 1e-12, k
        assert not np.any(res[True][k][:N, N:]) and not np.any(res[True][k][N:, :N])
    assert np.allclose(res[True][4][0], res[False][4][0], rtol=1e-12, atol=1e-14)
    assert np.allclose(res[True][4][1], res[False][4][1], rtol=1e-12, atol=1e-12)
    assert rel_fro(res[True][0], oracle.GrInt(F, S, G(), E, w)) < TOL
    assert rel_fro(res[True][1], oracle.GrLessInt(F, S, G(), E, w, -1)) < TOL
    # spin mixing present: the fast path must step aside (result = full path = oracle)
    C = np.zeros((N, N)); C[3, 7] = C[7, 3] = 0.05
    Fm = np.block([[Fa, C], [C, Fb]])
    assert I._spin_split(Fm, S, G()) is None
    assert rel_fro(GrInt(Fm, S, G(), E, w), oracle.GrInt(Fm, S, G(), E, w)) < TOL


def test_config_C5_shape(engine):
    """C5: spin-polarised 2 x 1000 block F/S (scf.py:177-180 layout), constant Sigma expanded with
    kron(I2, sigma) (transport.py:100), two energies of the qV = 0.5 V window at T = 300 K:
    GrLessInt(ind=-1) and the spin-'u' block transmission."""
    from gaunegf_amd.integrate import GrLessInt
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
    N, nc = 1000, 30
    Fa, Sa = random_system(N, 5); Fb, _ = random_system(N, 6)
    Z = np.zeros((N, N))
    F = np.block([[Fa, Z], [Z, Fb]]); S = np.kron(np.eye(2), Sa)
    inds, s1, s2 = const_sigma_pair(N, Sa, nc)
    sig = [np.kron(np.eye(2), s1), np.kron(np.eye(2), s2)]

    class G:
        def sigma(self, E, i): return sig[i]
        def sigmaTot(self, E): return sig[0] + sig[1]
    E, w = oracle.bias_window_grid(-0.25, 0.25, 512, 300.0)
    sub = np.array([100, 400])
    from gaunegf_amd.surfGTester import surfGTest
    got = GrLessInt(F, S, G(), E[sub], w[sub], -1)
    assert rel_fro(got, oracle.GrLessInt(F, S, G(), E[sub], w[sub], -1)) < TOL
    sc = SigmaCalculator(s1, s2)
    Et = np.array([-0.1, 0.2])
    T, Ts = calculate_transmission(F, S, sc, Et, spin='u')
    g1 = np.kron(np.eye(2), 1j * (s1 - s1.conj().T)); g2 = np.kron(np.eye(2), 1j * (s2 - s2.conj().T))
    for k, e in enumerate(Et):
        tot, comp = oracle.transmission_spin_block(e, F, S, sig[0] + sig[1], g1, g2)
        assert np.max(np.abs(Ts[k] - comp)) < TOL * max(1.0, np.max(np.abs(comp)))
        assert abs(T[k] - tot) < TOL * max(1.0, abs(tot))


def _gapped_system(N, seed):
    """Real symmetric F, S with the generalised spectrum eig(inv(S) F) in [-5, -3] u [3, 5]: a grid in (-1, 1) sits in a
    gap, G(E) is real up to the broadening."""
    rng = np.random.default_rng(seed)
    _, S = random_system(N, seed)
    S = np.real(S); S = (S + S.T) / 2
    L = np.linalg.cholesky(S)
    Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
    d = np.concatenate([rng.uniform(-5, -3, N // 2), rng.uniform(3, 5, N - N // 2)])
    F = L @ (Q * d) @ Q.T @ L.T
    return (F + F.T) / 2, S


@pytest.mark.parametrize("N,algo", [(300, 0), (300, 3), (300, 4), (650, 0), (650, 3)])
def test_imaginary_part_in_a_spectral_gap_windowed_inverse(engine, N, algo):
    """The Im-only bar on the WINDOWED inverse in the regime it is about: a system with a gap built into its spectrum,
    the grid inside the gap at eta = 1e-4, two contact orbitals per side with Gamma = 2e-3 -- the oracle's ||Im|| is
    4e-5 ||Re|| (asserted: <= 1e-3), and the imaginary part ALONE of GrInt and of G(E) meets 1e-8 through both
    window-kernel families (negf_set_inverse_algo 3: register strips, vector FMAs; 4: team kernels, 3M matrix-core
    updates whose rounding error in Im is relative to |Re|) and through the 3M column updates both share."""
    from gaunegf_amd.integrate import GrBatch, GrInt
    from gaunegf_amd.surfGTester import surfGTest
    F, S = _gapped_system(N, 77 + N)
    inds = [[0, 1], [N - 2, N - 1]]
    g_dev = surfGTest(F, S, inds, -1e-3j)
    g_ref = oracle.ConstSigma(F, S, inds, -1e-3j)
    E = np.linspace(-1.0, 1.0, 6) + 1e-4j
    w = np.full(6, 2.0 / 6)
    ref = oracle.GrInt(F, S, g_ref, E, w)
    assert np.linalg.norm(ref.imag) < 1e-3 * np.linalg.norm(ref.real)          # (the regime: Im three orders below Re)
    engine.set_inverse_algo(algo)
    try:
        got = GrInt(F, S, g_dev, E, w)
        G = GrBatch(F, S, g_dev, E[:2])
    finally:
        engine.set_inverse_algo(0)
    assert rel_fro(got.imag, ref.imag) < TOL, (N, algo, rel_fro(got.imag, ref.imag))
    assert rel_fro(got.real, ref.real) < TOL
    for k in range(2):
        r = oracle.gr_point(g_ref.sigmaTot(E[k]), E[k], F, S)
        assert np.linalg.norm(r.imag) < 1e-3 * np.linalg.norm(r.real)
        assert rel_fro(G[k].imag, r.imag) < TOL, (N, algo, k, rel_fro(G[k].imag, r.imag))


@pytest.mark.parametrize("N", [60, 120, 300])
def test_imaginary_part_of_the_real_axis_integral(engine, N):
    """The dense kernels form complex products from three real ones (3M: Im = S3 - S1 - S2), whose rounding error in the
    imaginary part is relative to |Re|, not to |Im|.  On a real-axis grid at eta = 1e-6 with weakly coupled contacts the
    density lives in Im G, below Re G (||Im|| / ||Re|| = 0.006 / 0.003 / 0.055 for N = 60 / 120 / 300): the imaginary part
    ALONE of GrInt and of G(E) must still meet the 1e-8 bar against the oracle (N = 60: the fused small-system kernel; 120:
    the single-workgroup blocked inverse; 300: the windowed inverse -- whose regime proper, three orders below Re, is
    test_imaginary_part_in_a_spectral_gap_windowed_inverse)."""
    from gaunegf_amd.integrate import GrBatch, GrInt
    from gaunegf_amd.surfGTester import surfGTest
    F, S = random_system(N, 31 + N)
    nc = max(2, N // 20)
    inds = [list(range(nc)), list(range(N - nc, N))]
    g_dev = surfGTest(F, S, inds, -1e-4j)
    g_ref = oracle.ConstSigma(F, S, inds, -1e-4j)
    E = np.linspace(-2.0, 2.0, 24) + 1e-6j
    w = np.full(24, 4.0 / 24)
    ref = oracle.GrInt(F, S, g_ref, E, w)
    got = GrInt(F, S, g_dev, E, w)
    assert np.linalg.norm(ref.imag) < (1e-2 if N < 300 else 1e-1) * np.linalg.norm(ref.real)     # (the regime: Im << Re)
    assert rel_fro(got.imag, ref.imag) < TOL
    G = GrBatch(F, S, g_dev, E[:3])
    for k in range(3):
        r = oracle.gr_point(g_ref.sigmaTot(E[k]), E[k], F, S)
        assert rel_fro(G[k].imag, r.imag) < TOL, k
