"""
Round-robin execution of the chain fixed points (include/negf.h, negf_set_chain_round_robin): a launch with more
fixed points than resident slots runs them in quanta of sweeps through a device-side queue.  A fixed point is a
sequence of sweeps on its own data (gauNEGF/surfG1D.py:271-288), so neither the quantum nor the number of slots may
change a bit of Sigma, of the sweep counts or of the convergence flags.
"""
import numpy as np
import pytest

from helpers import chain_lead, random_system

pytestmark = pytest.mark.gpu


def _leads(N, ncL, ncR, seed, eta):
    from gaunegf_amd.surfG1D import surfG
    F, S = random_system(N, seed)
    left = list(range(ncL)); right = list(range(N - ncR, N))
    aL = chain_lead(ncL, seed + 1); aR = chain_lead(ncR, seed + 2)
    kw = dict(taus=[aL[2].copy(), aR[2].copy()], staus=[aL[3].copy(), aR[3].copy()], alphas=[aL[0], aR[0]],
              aOverlaps=[aL[1], aR[1]], betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=eta)
    return lambda: surfG(F, S, [left, right], **kw)


@pytest.fixture
def rr(engine):
    engine.set_chain_cache(0)                      # every evaluation runs its fixed points
    yield engine
    engine.set_chain_round_robin(-1, 0)
    engine.set_chain_cache(512)


@pytest.mark.parametrize("ncL,ncR,eta,M", [(10, 10, 1e-3, 61), (50, 50, 1e-3, 14), (50, 40, 1e-3, 12), (19, 19, 1e-3, 20),
                                           (33, 35, 1e-3, 16), (64, 57, 2e-3, 8)])
def test_round_robin_changes_no_bit(rr, ncL, ncR, eta, M):
    """Free-running fixed points (lengths from tens of sweeps to the 2000 cap) on 3 ... 11 slots with quanta of
    1 ... 400 sweeps against the plain launch (one workgroup per fixed point): Sigma, counts and flags bit for bit.
    (Evaluated twice per provider: the plain launch of the second evaluation starts from a predicted order.)"""
    make = _leads(ncL + ncR + 9, ncL, ncR, 500 + ncL, eta)
    E = np.linspace(-1.6, 1.5, M) + 0.0j
    E[M // 2] += 0.2j                              # (one energy off the axis: a short fixed point)
    rr.set_chain_round_robin(0, 0)
    sig0, it0, cv0 = make().sigma_batch(E)
    assert it0.max() > 4 * it0.min()
    for quantum, slots in ((1, 3), (7, 5), (64, 11), (400, 4)):
        rr.set_chain_round_robin(quantum, slots)
        g = make()
        for rep in range(2):
            sig, it, cv = g.sigma_batch(E)
            assert np.array_equal(it, it0) and np.array_equal(cv, cv0), (quantum, slots, rep)
            assert np.array_equal(sig, sig0), (quantum, slots, rep)


def test_round_robin_fixed_trip_and_cache(rr):
    """force_iters (every fixed point the same length: all of them are set aside and resumed several times) and the
    g(E) cache filled by a round-robin launch: the hit equals the plain launch bit for bit."""
    make = _leads(131, 50, 50, 77, 1e-4)
    E = np.linspace(-1.0, 1.0, 9) + 0.0j
    rr.set_chain_round_robin(0, 0)
    g = make(); g.force_iters = 23
    sig0, it0, _ = g.sigma_batch(E)
    assert (it0 == 23).all()
    rr.set_chain_round_robin(5, 4)
    g = make(); g.force_iters = 23
    sig1, it1, _ = g.sigma_batch(E)
    assert np.array_equal(it1, it0) and np.array_equal(sig1, sig0)
    rr.set_chain_cache(512)
    rr.chain_cache_clear()
    g = make(); g.force_iters = 23
    sig2, it2, _ = g.sigma_batch(E)                 # miss: round robin, stores g
    sig3, it3, _ = g.sigma_batch(E)                 # hit
    st = rr.chain_cache_stats()
    assert st["hits"] >= 1
    assert np.array_equal(sig2, sig0) and np.array_equal(sig3, sig0) and np.array_equal(it3, it0)


def test_round_robin_through_the_integral(rr):
    """GrInt over a chain provider (the BASELINE C3 path) with the queue forced on a small grid equals the plain launch."""
    from gaunegf_amd.integrate import GrInt
    N, nc = 90, 20
    make = _leads(N, nc, nc, 901, 1e-3)
    E = np.linspace(-1.5, 1.5, 40); w = np.full(40, 3.0 / 40) + 0j
    rr.set_chain_round_robin(0, 0)
    g = make()
    ref = GrInt(g.F, g.S, g, E, w)
    rr.set_chain_round_robin(9, 6)
    g = make()
    out = GrInt(g.F, g.S, g, E, w)
    assert np.array_equal(out, ref)
