"""
1-D chain providers on the GPU: contacts of unequal size in one launch, and the context's g(E) cache
(include/negf.h, negf_set_chain_cache) -- a hit must equal a miss bit for bit.

Reference semantics: g(E) depends on the lead cell, eta and the stopping parameters only (gauNEGF/surfG1D.py:256-288);
setF refreshes the coupling blocks tau, not the lead (:319-329); Sigma = t g t^H (:366-372).
"""
import numpy as np
import pytest

import oracle
from helpers import chain_lead, random_system, rel_fro

pytestmark = pytest.mark.gpu


def _two_lead_system(N, ncL, ncR, seed, eta):
    from gaunegf_amd.surfG1D import surfG
    F, S = random_system(N, seed)
    left = list(range(ncL)); right = list(range(N - ncR, N))
    aL = chain_lead(ncL, seed + 1); aR = chain_lead(ncR, seed + 2)
    taus = [aL[2].copy(), aR[2].copy()]; staus = [aL[3].copy(), aR[3].copy()]
    kw = dict(taus=taus, staus=staus, alphas=[aL[0], aR[0]], aOverlaps=[aL[1], aR[1]],
              betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=eta)
    g_dev = surfG(F, S, [left, right], **kw)
    g_ref = oracle.Chain1DSigma(F, S, [left, right], taus, staus, [aL[0], aR[0]], [aL[1], aR[1]],
                                [aL[2], aR[2]], [aL[3], aR[3]], eta=eta)
    return F, S, g_dev, g_ref, kw


@pytest.mark.parametrize("ncL,ncR", [(50, 40), (40, 50), (35, 20), (19, 9), (51, 42), (50, 48), (18, 17), (34, 32),
                                      (64, 3), (50, 49)])
def test_chain1d_unequal_contacts_fixed_trip(engine, ncL, ncR):
    """Contacts of unequal size share a launch and one padded work matrix whose pitch class follows the LARGER one.
    The remainder-strip classes (19, 35, 51) assume that every contact reaches into the strip; a launch whose smaller
    contact does not -- (50, 40), (35, 20), (19, 9), (51, 42), (50, 48), (18, 16+1 = 17 does), (34, 32) -- must take
    the guarded class.  Fixed trip count -> the iterate itself agrees with the oracle for BOTH contacts; 42 sits just
    below a k-step boundary, 48 and 32 exactly on the strip base."""
    N = ncL + ncR + 11
    F, S, g_dev, g_ref, _ = _two_lead_system(N, ncL, ncR, 300 + ncL + ncR, 1e-4)
    g_dev.force_iters = 40; g_ref.force_iters = 40
    E = np.array([0.3, -0.8, 0.1 + 0.2j])
    for i in (0, 1):
        sig, it, cv = g_dev.sigma_batch(E, i)
        assert np.all(it == 40)
        for k, e in enumerate(E):
            assert rel_fro(sig[k], g_ref.sigma(e, i)) < 1e-10, (ncL, ncR, e, i)
    sig, _, _ = g_dev.sigma_batch(E)
    for k, e in enumerate(E):
        assert rel_fro(sig[k], g_ref.sigmaTot(e)) < 1e-10
    # free running, both contacts: counts within +-1 of the oracle's
    g_dev.force_iters = -1; g_dev._version += 1; g_ref.force_iters = None
    g_ref.eta = g_dev.eta = 1e-2; g_dev._version += 1
    sig, it, cv = g_dev.sigma_batch(E[:2])
    for k, e in enumerate(E[:2]):
        ref = g_ref.sigmaTot(e)
        for c in (0, 1):
            assert abs(int(it[k, c]) - g_ref.last_iters[(complex(e), c)][0]) <= 1, (ncL, ncR, e, c)
        assert rel_fro(sig[k], ref) < 1e-4


@pytest.fixture
def cache(engine):
    engine.set_chain_cache(0)            # drop whatever earlier tests left
    engine.set_chain_cache(512)
    yield engine
    engine.set_chain_cache(0)
    engine.set_chain_cache(512)


def _stats(engine):
    s = engine.chain_cache_stats()
    return s["hits"], s["misses"]


@pytest.mark.parametrize("nc", [10, 33, 50, 64])
def test_gcache_hit_equals_miss_bit_for_bit(cache, nc):
    """The same grid twice: the second launch is a hit (it only forms Sigma = t g t^H from the stored iterate) and
    must return Sigma, the sweep counts and the convergence flags of the first, bit for bit -- production stopping
    rule, jobs of different length, every kernel class (n_c = 10: one tile; 33: three; 50: remainder strip + global
    scratch for the old iterate; 64: four tiles)."""
    eng = cache
    F, S, g, _, _ = _two_lead_system(2 * nc + 9, nc, nc, 500 + nc, 1e-3)
    E = np.concatenate([np.linspace(-1.5, 1.5, 21), [0.2 + 0.1j]])
    h0, m0 = _stats(eng)
    sig, it, cv = g.sigma_batch(E)
    h1, m1 = _stats(eng)
    assert (h1 - h0, m1 - m0) == (0, 1)
    sig2, it2, cv2 = g.sigma_batch(E)
    h2, m2 = _stats(eng)
    assert (h2 - h1, m2 - m1) == (1, 0)
    assert np.array_equal(sig, sig2) and np.array_equal(it, it2) and np.array_equal(cv, cv2)
    assert it.max() > it.min()
    # one contact's Sigma from the same entry, and against a cold evaluation
    s0, it0, _ = g.sigma_batch(E, 0)
    assert _stats(eng)[0] == h2 + 1 and np.array_equal(it0, it)
    eng.set_chain_cache(0)
    s0c, it0c, _ = g.sigma_batch(E, 0)
    sigc, itc, cvc = g.sigma_batch(E)
    eng.set_chain_cache(512)
    assert np.array_equal(s0, s0c) and np.array_equal(sig, sigc) and np.array_equal(it, itc) and np.array_equal(cv, cvc)
    assert _stats(eng) == (h2 + 1, m2)                      # a switched-off cache counts nothing


def test_gcache_survives_setF_and_serves_the_t_identity_variant(cache):
    """tau changes (setF), the lead does not: the provider is rebuilt, its launches hit the entries of its predecessor
    and form Sigma with the NEW tau -- equal, bit for bit, to a cold evaluation of the new provider and to the oracle
    at a fixed trip count.  surfG.g() (t = I) reads the same entries."""
    from gaunegf_amd.surfG1D import surfG
    eng = cache
    N = 30
    F, S = random_system(N, 8)
    inds = [[0, 1, 2, 3, 4], [25, 26, 27, 28, 29]]
    conn = [[5, 6, 7, 8, 9], [20, 21, 22, 23, 24]]
    g = surfG(F, S, inds, taus=conn, eta=1e-3)
    E = np.linspace(-1.0, 1.0, 9)
    sigA, itA, _ = g.sigma_batch(E)
    h0, m0 = _stats(eng)
    rng = np.random.default_rng(1)
    D = rng.standard_normal((N, N)); F2 = F + 0.05 * (D + D.T)
    g.setF(F2, 0.0, 0.0)
    sigB, itB, cvB = g.sigma_batch(E)                      # new provider, new tau, cached g
    assert _stats(eng) == (h0 + 1, m0)
    assert np.array_equal(itA, itB) and not np.array_equal(sigA, sigB)
    eng.set_chain_cache(0)
    sigC, itC, cvC = g.sigma_batch(E)                      # cold
    eng.set_chain_cache(512)
    assert np.array_equal(sigB, sigC) and np.array_equal(itB, itC) and np.array_equal(cvB, cvC)
    # t = I: the surface Green's function itself, for a single energy (its own one-point entry, then a hit)
    g0 = g.g(E[3], 0)
    h1, m1 = _stats(eng)
    g1 = g.g(E[3], 1)
    assert _stats(eng) == (h1 + 1, m1) and not np.array_equal(g0, g1)
    ref, cnt, _ = oracle.chain1d_g(E[3], g.aList[0], g.aSList[0], g.bList[0], g.bSList[0], 1e-3)
    assert rel_fro(g0, ref) < 1e-4
    # fixed trip count against the oracle, through a hit
    g.force_iters = 30; g._version += 1
    s1, _, _ = g.sigma_batch(E[:3]); s2, _, _ = g.sigma_batch(E[:3])
    r = oracle.Chain1DSigma(g.F, g.S, inds, g.tauList, g.stauList, g.aList, g.aSList, g.bList, g.bSList, eta=1e-3)
    r.force_iters = 30
    assert np.array_equal(s1, s2)
    for k in range(3):
        assert rel_fro(s2[k], r.sigmaTot(E[k])) < 1e-10


def test_gcache_invalidation_eviction_two_providers(cache):
    """A new lead cell (setContacts), another eta, conv or grid is a miss; entries are evicted least recently used;
    two providers with different leads keep their own entries."""
    eng = cache
    nc = 12
    F, S, gA, _, kwA = _two_lead_system(40, nc, nc, 700, 1e-3)
    _, _, gB, _, _ = _two_lead_system(40, nc, nc, 900, 1e-3)
    E1 = np.linspace(-1, 1, 7); E2 = np.linspace(-1, 1, 8); E3 = E1 + 1e-9
    a1, ia1, _ = gA.sigma_batch(E1); b1, ib1, _ = gB.sigma_batch(E1)
    h, m = _stats(eng)
    a1h, _, _ = gA.sigma_batch(E1); b1h, _, _ = gB.sigma_batch(E1)
    assert _stats(eng) == (h + 2, m) and np.array_equal(a1, a1h) and np.array_equal(b1, b1h) and not np.array_equal(a1, b1)
    # other grids / parameters: misses
    gA.sigma_batch(E2); gA.sigma_batch(E3)
    assert _stats(eng) == (h + 2, m + 2)
    gA.eta = 2e-3; gA._version += 1
    gA.sigma_batch(E1)
    assert _stats(eng) == (h + 2, m + 3)
    gA.eta = 1e-3; gA._version += 1
    from gaunegf_amd.config import SURFACE_GREEN_CONVERGENCE
    hh = gA._negf_lower(eng, conv=10 * SURFACE_GREEN_CONVERGENCE)
    eng.sigma_eval(hh, None, E1, 2)
    assert _stats(eng) == (h + 2, m + 4)
    # a new lead cell: miss, and the result is that of a fresh object
    al = [a + 0.01 * np.eye(nc) for a in kwA["alphas"]]
    gA.setContacts(al, kwA["aOverlaps"], kwA["betas"], kwA["bOverlaps"])
    a2, ia2, _ = gA.sigma_batch(E1)
    assert _stats(eng) == (h + 2, m + 5) and not np.array_equal(a2, a1)
    eng.set_chain_cache(0)
    a2c, ia2c, _ = gA.sigma_batch(E1)
    assert np.array_equal(a2, a2c) and np.array_equal(ia2, ia2c)
    # eviction: two entries only
    eng.set_chain_cache(2)
    h, m = _stats(eng)
    gB.sigma_batch(E1); gB.sigma_batch(E2); gB.sigma_batch(E1)            # E1 hit, E2 now least recently used
    assert _stats(eng) == (h + 1, m + 2)
    gB.sigma_batch(E3)                                                    # evicts E2
    x1, _, _ = gB.sigma_batch(E1)
    assert _stats(eng) == (h + 2, m + 3)
    gB.sigma_batch(E2)                                                    # gone: a miss again
    assert _stats(eng) == (h + 2, m + 4) and eng.chain_cache_stats()["entries"] == 2
    assert np.array_equal(x1, b1)


def test_gcache_through_the_integrals_and_the_dev_entry_points(cache):
    """GrInt / GrLessInt / transmission on one grid: the first fills the entry, the others hit it; cached == cold bit
    for bit, also through the device-pointer entry points (which download their energy list to form the key)."""
    from gaunegf_amd.integrate import GrInt, GrLessInt
    from gaunegf_amd.transport import SigmaCalculator, calculate_transmission
    eng = cache
    nc = 18
    F, S, g, g_ref, _ = _two_lead_system(70, nc, nc, 1100, 1e-3)
    E, w = oracle.bias_window_grid(-0.3, 0.3, 24, 300.0)

    def run():
        return (GrInt(F, S, g, E, w), GrLessInt(F, S, g, E, w, -1), GrLessInt(F, S, g, E, w, None),
                calculate_transmission(F, S, SigmaCalculator(g), np.real(E)))
    eng.set_chain_cache(0)
    cold = run()
    eng.set_chain_cache(512)
    h, m = _stats(eng)
    warm = run()
    h1, m1 = _stats(eng)
    assert m1 - m == 1 and h1 - h >= 3, (h1 - h, m1 - m)
    for a, b in zip(cold, warm):
        assert np.array_equal(a, b)
    assert rel_fro(warm[0], oracle.GrInt(F, S, g_ref, E, w)) < 1e-3
    # device-pointer entry point (buffers through the HIP runtime the library itself is linked against)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]

    def dev_buf(a):
        a = np.ascontiguousarray(a, dtype=np.complex128)
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), a.nbytes) == 0
        assert hip.hipMemcpy(ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0          # host -> device
        return ptr
    eng.set_system(F, S)
    hnd = g._negf_lower(eng)
    Ed, wd, outd = dev_buf(E), dev_buf(w), dev_buf(np.zeros((70, 70), dtype=np.complex128))
    try:
        h, m = _stats(eng)
        eng.gr_int_dev(hnd, E.size, Ed.value, wd.value, outd.value); eng.sync()
        assert _stats(eng) == (h + 1, m)
        res = np.zeros((70, 70), dtype=np.complex128)
        assert hip.hipMemcpy(res.ctypes.data_as(C.c_void_p), outd, res.nbytes, 2) == 0      # device -> host
        assert np.array_equal(res, cold[0])
    finally:
        for b in (Ed, wd, outd):
            hip.hipFree(b)


def test_gcache_byte_budget(cache):
    """The cache is bounded in bytes as well as in entries: with room for two of three equally large launches the least
    recently used one goes, and a launch larger than the whole budget is simply not cached."""
    eng = cache
    nc = 16
    F, S, g, _, _ = _two_lead_system(50, nc, nc, 1300, 1e-3)
    grids = [np.linspace(-1, 1, 40) + 0.01 * k for k in range(3)]
    per = 40 * 2 * nc * nc * 16 + 2 * 40 * 2 * 4                # bytes of one entry: g blocks + counts and flags
    eng.set_chain_cache(max_bytes=2 * per + 64)
    try:
        h, m = _stats(eng)
        ref = [g.sigma_batch(E)[0] for E in grids]               # third insertion evicts the first
        assert _stats(eng) == (h, m + 3) and eng.chain_cache_stats()["entries"] == 2
        assert np.array_equal(g.sigma_batch(grids[2])[0], ref[2]) and np.array_equal(g.sigma_batch(grids[1])[0], ref[1])
        assert _stats(eng) == (h + 2, m + 3)
        assert np.array_equal(g.sigma_batch(grids[0])[0], ref[0])
        assert _stats(eng) == (h + 2, m + 4)
        eng.set_chain_cache(max_bytes=per // 2)                  # nothing of this size fits any more
        eng.chain_cache_clear()
        g.sigma_batch(grids[0]); g.sigma_batch(grids[0])
        assert eng.chain_cache_stats()["entries"] == 0
    finally:
        eng.set_chain_cache(max_bytes=8 << 30)
