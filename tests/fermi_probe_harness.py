"""
Analytic "engine" behind the Fermi-level searches and the integration-limit fitting
(reference gauNEGF/density.py:821-1515).

Those functions are pure numpy host logic around a handful of callables -- densityComplexN /
densityComplex / densityRealN / densityReal / densityGridN, _compute_dos_at_energy, g.setF,
g.sigmaTot.  Both sides are run against the SAME spies from this file:

  * tests/golden/make_golden.py executes the reference's own function bodies (AST-extracted,
    unmodified) with the spies bound to those names and commits the recorded probe sequences
    (tests/golden/ref_fermi_search.npz);
  * tests/test_fermi_search_golden.py runs gaunegf_amd.density with the same spies bound and
    compares probe for probe (bit-exact chemical potentials) and the returned levels.

The spies model a set of Lorentzian levels: occupation n_i(mu) = 1/2 + atan((mu - e_i)/gamma)/pi,
DOS(E) = sum gamma/pi/((E - e_i)^2 + gamma^2); the fixed-N integrals carry a factor (1 - 1/N^2) so
that the grid-doubling loops have something to converge.
"""
import numpy as np

EPS = np.array([-3.0, -1.2, -0.4, 0.3, 1.1, 2.5])
GAMMA = 0.15


def occ(mu):
    return 0.5 + np.arctan((float(mu) - EPS) / GAMMA) / np.pi


def dos(E):
    return float(np.sum(GAMMA / np.pi / ((float(np.real(E)) - EPS) ** 2 + GAMMA ** 2)))


class ProbeG:
    """Duck-typed surface-Green's-function object: F, S, setF, sigmaTot."""
    def __init__(self, rec):
        self.F = np.diag(EPS)
        self.S = np.eye(len(EPS))
        self.rec = rec

    def setF(self, F, mu1, mu2):
        self.rec.append(("setF", float(mu1), float(mu2), 0.0, 0.0))

    def sigmaTot(self, E, conv=None):
        return np.zeros((len(EPS), len(EPS)), dtype=complex)


def make_spies(rec):
    """name -> callable, with the reference's positional/keyword signatures."""
    def densityComplexN(F, S, g, Emin, mu, N=100, T=300, showText=True, method='ant'):
        rec.append(("cplxN", float(Emin), float(mu), float(N), float(T)))
        return np.diag((occ(mu) - occ(Emin)) * (1.0 - 1.0 / float(N) ** 2)).astype(complex)

    def densityComplex(F, S, g, Emin, mu, tol=1e-3, T=300, debug=False):
        rec.append(("cplx", float(Emin), float(mu), float(tol), float(T)))
        return np.diag(occ(mu) - occ(Emin)).astype(complex)

    def densityRealN(F, S, g, Emin, mu, N=100, T=300, showText=True):
        rec.append(("realN", float(Emin), float(mu), float(N), float(T)))
        return np.diag((occ(mu) - occ(Emin)) * (1.0 - 1.0 / float(N) ** 2)).astype(complex)

    def densityReal(F, S, g, Emin, mu, tol=1e-3, T=300, maxN=1000, debug=False):
        rec.append(("real", float(Emin), float(mu), float(tol), float(T)))
        return np.diag(occ(mu) - occ(Emin)).astype(complex)

    def densityGridN(F, S, g, mu1, mu2, ind=None, N=100, T=300, showText=True):
        rec.append(("gridN", float(mu1), float(mu2), float(N), float(-99 if ind is None else ind)))
        return np.diag(0.5 * (occ(mu2) - occ(mu1)) * (1.0 - 1.0 / float(N) ** 2)).astype(complex)

    def _compute_dos_at_energy(E, F, S, sigma_total):
        rec.append(("dos", float(np.real(E)), 0.0, 0.0, 0.0))
        return dos(E)

    return dict(densityComplexN=densityComplexN, densityComplex=densityComplex, densityRealN=densityRealN,
                densityReal=densityReal, densityGridN=densityGridN, _compute_dos_at_energy=_compute_dos_at_energy)


# (tag, function name, how to call it given (fn, g)) -- the cases both sides run
NE = 2.7
CASES = [
    ("emin", "calcEmin", lambda fn, g: fn(g.F, g.S, g, 1e-3, 40)),
    ("fit", "integralFit", lambda fn, g: fn(g.F, g.S, g, 0.2, -1e6, 1e-4, 300.0, 300)),
    ("fitnegf", "integralFitNEGF", lambda fn, g: fn(g.F, g.S, g, 0.1, 0.4, -1e6, 1e-4, 300.0, 400)),
    ("bisect_N", "calcFermiBisect", lambda fn, g: fn(g, NE, -20.0, 0.4, 32, 1e-3, 1e-6, 40, 300.0)),
    ("bisect_adapt", "calcFermiBisect", lambda fn, g: fn(g, NE, -20.0, -0.9, None, 1e-3, 1e-5, 40, 0.0)),
    ("bisect_bounds", "calcFermiBisect", lambda fn, g: fn(g, NE, -20.0, 0.0, 64, 1e-3, 1e-6, 40, 300.0, 1.0, -1.0)),
    ("secant_N", "calcFermiSecant", lambda fn, g: fn(g, NE, -20.0, 0.4, 32, 1e-3, 1e-6, 30, 300.0)),
    ("secant_adapt", "calcFermiSecant", lambda fn, g: fn(g, NE, -20.0, -0.6, None, 1e-3, 1e-7, 30, 0.0)),
    ("muller_N", "calcFermiMuller", lambda fn, g: fn(g, NE, -20.0, 0.4, 32, 1e-3, 1e-6, 30, 300.0)),
    ("muller_adapt", "calcFermiMuller", lambda fn, g: fn(g, 4.1, -20.0, 0.2, None, 1e-3, 1e-7, 30, 0.0)),
    ("poly_N", "calcFermiPolyFit", lambda fn, g: fn(g, NE, -20.0, 0.4, 32, 1e-3, 1e-6, 30, 300.0)),
    ("poly_adapt", "calcFermiPolyFit", lambda fn, g: fn(g, 4.1, -20.0, 0.2, None, 1e-3, 1e-7, 30, 0.0)),
]

KINDS = ["setF", "cplxN", "cplx", "realN", "real", "gridN", "dos"]


def pack(rec):
    """probe list -> float array [n, 5] (kind index, four numbers)."""
    return np.array([[KINDS.index(r[0])] + list(r[1:]) for r in rec], dtype=float).reshape(-1, 5)


def scalars(ret):
    """the scalar part of a search function's return value (matrices dropped, None -> nan)."""
    if not isinstance(ret, tuple):
        ret = (ret,)
    out = []
    for v in ret:
        if v is None:
            out.append(np.nan)
        elif np.ndim(v) == 0:
            out.append(float(np.real(v)))
    return np.array(out)
