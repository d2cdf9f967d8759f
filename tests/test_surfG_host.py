"""
Host-side construction patterns of the 1-D chain provider (gauNEGF/surfG1D.py:83-221) and setF
(:297-342): what the drop-in extracts from F / S must be what the reference's constructor extracts
(its formulas are restated here, line by line, on plain numpy index expressions).  No GPU needed:
nothing is evaluated, only the operands that would be lowered to the device are checked.
"""
import numpy as np
import pytest

from helpers import chain_lead, random_system


def _system(N=18, seed=3):
    F, S = random_system(N, seed)
    return F, S


def test_pattern_a_indices_only():
    """surfG1D.py:131-139, 145-147, 193-215: everything from F / S; default taus = [inds[-1], inds[0]]."""
    from gaunegf_amd.surfG1D import surfG
    F, S = _system()
    inds = [[0, 1, 2], [15, 16, 17]]
    g = surfG(F, S, inds)
    assert g.tauFromFock and g.contactFromFock and g.num_contacts == 2
    assert [list(t) for t in g.tauInds] == [inds[-1], inds[0]]
    t = [np.array(inds[-1]), np.array(inds[0])]
    exp_tau = [F[np.ix_(t[0], inds[0])], F[np.ix_(t[1], inds[-1])]]
    exp_stau = [S[np.ix_(t[0], inds[0])], S[np.ix_(t[1], inds[-1])]]
    for k in range(2):
        assert np.array_equal(g.tauList[k], exp_tau[k]) and np.array_equal(g.stauList[k], exp_stau[k])
        assert np.array_equal(g.aList[k], F[np.ix_(inds[k], inds[k])])
        assert np.array_equal(g.aSList[k], S[np.ix_(inds[k], inds[k])])
        assert np.array_equal(g.bList[k], exp_tau[k]) and np.array_equal(g.bSList[k], exp_stau[k])
    assert np.allclose(g.X @ g.X @ S, np.eye(len(S)), atol=1e-10)          # X = S^(-1/2)


def test_pattern_a_explicit_connection_indices_and_pattern_b_matrices():
    from gaunegf_amd.surfG1D import surfG
    F, S = _system()
    inds = [[0, 1, 2], [15, 16, 17]]
    conn = [[3, 4, 5], [12, 13, 14]]                       # the device atoms each contact couples to
    g = surfG(F, S, inds, taus=conn)
    assert g.tauFromFock and g.contactFromFock
    assert np.array_equal(g.tauList[0], F[np.ix_(conn[0], inds[0])])
    assert np.array_equal(g.tauList[1], F[np.ix_(conn[1], inds[1])])
    assert np.array_equal(g.bSList[1], S[np.ix_(conn[1], inds[1])])
    # pattern (b): coupling matrices given, lead cell still from F / S (surfG1D.py:140-143, 145-147)
    tau = [np.full((3, 3), 0.1), np.full((3, 3), -0.2)]; stau = [np.zeros((3, 3)), np.eye(3) * 0.01]
    gb = surfG(F, S, inds, taus=tau, staus=stau)
    assert not gb.tauFromFock and gb.contactFromFock
    for k in range(2):
        assert np.array_equal(gb.tauList[k], tau[k]) and np.array_equal(gb.stauList[k], stau[k])
        assert np.array_equal(gb.aList[k], F[np.ix_(inds[k], inds[k])])
        assert np.array_equal(gb.bList[k], tau[k]) and np.array_equal(gb.bSList[k], stau[k])


def test_pattern_c_fully_specified_and_setF_shifts():
    """surfG1D.py:148-152: fully specified lead; setF (:330-342): the first call records the contact
    chemical potentials, later calls shift alpha by d mu * I and beta by d mu * S_beta (the reference's
    intent; its own code calls .at on a Python list there and would raise)."""
    from gaunegf_amd.surfG1D import surfG
    F, S = _system()
    inds = [[0, 1, 2], [15, 16, 17]]
    aL = chain_lead(3, 1); aR = chain_lead(3, 2)
    kw = dict(taus=[aL[2], aR[2]], staus=[aL[3], aR[3]], alphas=[aL[0], aR[0]], aOverlaps=[aL[1], aR[1]],
              betas=[aL[2], aR[2]], bOverlaps=[aL[3], aR[3]], eta=1e-4)
    g = surfG(F, S, inds, **kw)
    assert not g.tauFromFock and not g.contactFromFock and g.fermiList == [None, None]
    v0 = g._version
    F2 = F + 0.01
    g.setF(F2, 0.1, -0.1)
    assert np.array_equal(g.F, F2) and g.fermiList == [0.1, -0.1] and g._version > v0
    assert np.array_equal(g.aList[0], aL[0]) and np.array_equal(g.bList[1], aR[2])     # first call: no shift
    g.setF(F2, 0.3, -0.1)
    assert np.allclose(g.aList[0], aL[0] + 0.2 * np.eye(3)) and np.allclose(g.bList[0], aL[2] + 0.2 * aL[3])
    assert np.array_equal(g.aList[1], aR[0]) and g.fermiList == [0.3, -0.1]
    g.setF(F2, None, None)                                                             # None: untouched
    assert g.fermiList == [0.3, -0.1]


def test_setF_tau_from_fock_refreshes_coupling_not_lead():
    """surfG1D.py:319-329: with the coupling taken from F, a new F first copies the diagonal blocks of the
    connection atoms onto the contact atoms, then refreshes tau / S_tau; the lead cell (alpha, beta)
    extracted at construction is NOT refreshed (setContacts is not called there)."""
    from gaunegf_amd.surfG1D import surfG
    F, S = _system()
    inds = [[0, 1, 2], [15, 16, 17]]
    conn = [[3, 4, 5], [12, 13, 14]]
    g = surfG(F, S, inds, taus=conn)
    a_before = [a.copy() for a in g.aList]; b_before = [b.copy() for b in g.bList]
    rng = np.random.default_rng(0)
    D = rng.standard_normal(F.shape); F2 = F + 0.05 * (D + D.T)
    g.setF(F2, 0.0, 0.0)
    exp = F2.copy()
    exp[np.ix_(inds[0], inds[0])] = F2[np.ix_(conn[0], conn[0])]
    exp[np.ix_(inds[1], inds[1])] = F2[np.ix_(conn[1], conn[1])]
    assert np.array_equal(g.F, exp)
    assert np.array_equal(g.tauList[0], exp[np.ix_(conn[0], inds[0])])
    assert np.array_equal(g.tauList[1], exp[np.ix_(conn[1], inds[1])])
    assert np.array_equal(g.stauList[0], S[np.ix_(conn[0], inds[0])])
    for k in range(2):
        assert np.array_equal(g.aList[k], a_before[k]) and np.array_equal(g.bList[k], b_before[k])


def test_coupling_block_shape_is_checked_at_lowering():
    """t g t^H is added at ix_(inds, inds) (surfG1D.py:372): a coupling block that is not n_c x n_c cannot
    be lowered; the error names the contact."""
    from gaunegf_amd.surfG1D import surfG
    F, S = _system()
    g = surfG(F, S, [[0, 1, 2], [15, 16, 17]], taus=[[3, 4], [13, 14]])

    class FakeEngine:
        generation = 0
        def sigma_chain1d(self, *a, **k): raise AssertionError("must not be reached")
    with pytest.raises(ValueError, match="contact 0"):
        g._negf_lower(FakeEngine())
