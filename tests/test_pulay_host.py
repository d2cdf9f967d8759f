"""Pulay/DIIS mixing (scf.py:597-661 -> NEGFE.PMix) on the host: the mixing algebra needs no GPU."""
import numpy as np


def _negfe(n=6):
    from gaunegf_amd.scfE import NEGFE

    class G:                                   # the contact is not evaluated by PMix
        def setF(self, *a): pass
    obj = NEGFE(np.diag(np.arange(n, dtype=float)), np.eye(n), G(), ne=4)
    return obj


def test_damping_step_and_history_shift(capsys):
    s = _negfe()
    n = len(s.F)
    s.P_in = np.zeros((n, n), dtype=complex)
    s._init_pulay(3)
    rng = np.random.default_rng(0)
    P1 = rng.standard_normal((n, n)) + 0j
    s.P = P1.copy()
    rms, mx = s.PMix(0.25, False)
    assert np.allclose(s.P, 0.25 * P1)                                  # Pback + damping (P - Pback), Pback = 0
    assert np.isclose(mx, np.max(np.abs(np.diag(P1)))) and np.isclose(rms, np.sqrt(np.mean(np.diag(P1) ** 2)))
    assert np.allclose(s.DPList[0], P1) and np.allclose(s.pList[0], 0.25 * P1)
    assert np.isclose(s.nelec, 2 * np.real(np.trace(s.P @ s.S)))        # restricted: doubled (scf.py:261-265)
    prev = s.P.copy()
    s.P = prev + 0.1
    s.PMix(0.5, False)
    assert np.allclose(s.pList[1], 0.25 * P1) and np.allclose(s.DPList[1], P1)    # history shifted down
    assert np.allclose(s.P, prev + 0.05)


def test_diis_solves_a_linear_fixed_point_exactly(capsys):
    """For a linear map P -> A P + b whose residuals span a space of dimension <= nPulay - 1, the DIIS
    combination of the history (coefficients summing to one, minimising the residual norm) IS the fixed point."""
    s = _negfe(4)
    n = 4
    fixed = np.diag([0.3, -0.2, 0.5, 0.1]).astype(complex)
    lam = np.array([0.6, -0.4, 0.6, -0.4])                              # two distinct eigenvalues -> residuals in a 2-d space

    def step(P):                                                        # output density for input P
        return fixed + lam[:, None] * (P - fixed)
    s.P_in = np.zeros((n, n), dtype=complex) + 0.05
    s._init_pulay(3)
    for it in range(3):
        s.P = step(s.P_in)
        s.PMix(1.0, it == 2)                                            # damping 1: pList holds the map's outputs
    assert np.max(np.abs(s.P - fixed)) < 1e-10
    coeff = np.linalg.solve(s.pMat, s.pB)[:-1]
    assert abs(np.sum(coeff) - 1.0) < 1e-10
