"""
Density-matrix front-ends: energy grids, weights and prefactors on the host, every
Green's function on the GPU (drop-in for the energy-dependent part of
gauNEGF/density.py).

The grid bookkeeping must be bit-exact, so each builder
below performs the same floating-point operations in the same order as the
reference lines it cites; tests/test_grids.py compares them bit-for-bit against
vectors captured from the reference's own functions (tests/golden/).

    grid builder            reference                     integral        prefactor
    real_axis_grid          densityRealN  :418-427        GrInt           -Im(.)/pi
    bias_window_grid        densityGridN  :519-534        GrLessInt       (.)/(2 pi)
    contour_grid            densityComplexN :697-722      GrInt           +Im(.)/pi
    broadening_grid         densityComplexN :730-742      GrInt           (added to the contour)
"""
import numpy as np
from scipy.special import roots_legendre

from .config import (TEMPERATURE, ADAPTIVE_INTEGRATION_TOL, N_KT, MAX_CYCLES, MAX_GRID_POINTS)
from .integrate import GrInt, GrLessInt

har_to_eV = 27.211386   # eV/Hartree
kB = 8.617e-5           # eV/Kelvin


# ------------------------------------------------------------------ helpers
def fermi(E, mu, T):
    """Fermi-Dirac occupation (density.py:64-86).  At T == 0 the step ``(E<=mu)*1`` is
    evaluated with numpy's ordering, which for complex E is lexicographic."""
    kT = kB * T
    if kT == 0:
        return (E <= mu) * 1
    return 1 / (np.exp((E - mu) / kT) + 1)


def getANTPoints(N):
    """ANT.Gaussian-style modified Gauss-Chebyshev rule on [-1,1] (density.py:88-119):
    nested for N = 2*3^k, always an even number of points (+x then -x)."""
    k = np.arange(1, N + 1, 2)
    theta = k * np.pi / (2 * N)
    xs = np.sin(theta)
    xcc = np.cos(theta)
    x = 1.0 + 0.21220659078919378103 * xs * xcc * (3 + 2 * xs * xs) - k / (N)
    w = xs ** 4 * 16.0 / (3 * (N))
    return np.concatenate((x, -1 * x)), np.concatenate((w, w))


def integratePointsAdaptiveANT(computePoint, tol=ADAPTIVE_INTEGRATION_TOL, maxN=MAX_GRID_POINTS, debug=False):
    """Adaptive nested quadrature (density.py:211-273): levels N = 2, 6, 18, ...; each
    level sends ONLY its new nodes to ``computePoint(x, w)`` and rescales the running
    value by the nested-weight ratio; stops when max|dP| < tol or N would exceed maxN."""
    prev_x = prev_sumW = P = new_P = None
    N = 2
    maxDP = 1e10
    while N <= maxN:
        x, w = getANTPoints(N)
        if prev_x is None:
            P = computePoint(x[0:2], w[0:2])
        else:
            old_mask = np.isin(np.round(x, 14), np.round(prev_x, 14))
            assert int(old_mask.sum()) == prev_x.size, "Old nodes mismatch"
            ratio = float(np.sum(w[old_mask]) / prev_sumW)
            new_mask = ~old_mask
            new_P = P * ratio
            new_P += computePoint(x[new_mask], w[new_mask])
            maxDP = np.max(np.abs(new_P - P))
            if debug:
                direct = computePoint(x, w)
                print(f"N={N}, nested-weight ratio ~ {ratio:.3f}, maxDP={maxDP:.3e}")
                print(f"Direct Calculation: N={N}, maxDP={np.max(np.abs(direct - P)):.3e}, "
                      f"maxDiff={np.max(np.abs(direct - new_P)):.3e}")
            P = new_P.copy()
            if maxDP < tol:
                print(f'Adaptive integration converged to {maxDP:.3e} in {N} points.')
                return new_P
        prev_x = x
        prev_sumW = float(np.sum(w))
        N *= 3
    N /= 3
    print(f'Adaptive integration reached full grid ({N} points), final error {maxDP:.3e}')
    return new_P


# ------------------------------------------------------------ grid builders
def real_axis_grid(Emin, mu, N, T):
    """(Elist, weights) of densityRealN (density.py:418-427)."""
    kT = kB * T
    Emax = mu + N_KT * kT
    mid = (Emax - Emin) / 2
    x, w = roots_legendre(N)
    x = np.real(x)
    Elist = mid * (x + 1) + Emin
    weights = mid * w * fermi(Elist, mu, T)
    return Elist, weights


def _window(mu1, mu2, T):
    kT = kB * T
    muLo = min(mu1, mu2)
    muHi = max(mu1, mu2)
    dInt = np.sign(mu2 - mu1)
    Emax = muHi + N_KT * kT
    Emin = muLo - N_KT * kT
    return muLo, muHi, dInt, Emin, Emax


def bias_window_grid(mu1, mu2, N, T):
    """(energies, weights) of densityGridN (density.py:519-534)."""
    muLo, muHi, dInt, Emin, Emax = _window(mu1, mu2, T)
    mid = (Emax - Emin) / 2
    x, w = roots_legendre(N)
    x = np.real(x)
    energies = mid * (x + 1) + Emin
    dfermi = fermi(energies, muHi, T) - fermi(energies, muLo, T)
    weights = mid * w * dfermi * dInt
    return energies, weights


def _contour(Emin, mu, T):
    broadening = 10 * kB * T
    Emax = mu - broadening
    center = (Emin + Emax) / 2
    r = (Emax - Emin) / 2
    return broadening, center, r


def _rule(N, method):
    if method == 'legendre':
        return roots_legendre(N)
    if method == 'chebyshev':
        k = np.arange(1, N + 1)
        x = np.cos(k * np.pi / (N + 1))
        w = (np.pi / (N + 1)) * (np.sin(k * np.pi / (N + 1)) ** 2) / np.sqrt(1 - (x ** 2))
        return x, w
    if method == 'ant':
        return getANTPoints(N)
    return np.linspace(-1, 1, N), 2 * np.ones(N) / N      # midpoint rule


def contour_grid(Emin, mu, N, T, method='ant'):
    """(Elist, weights) on the upper semicircle of densityComplexN (density.py:697-722)."""
    _, center, r = _contour(Emin, mu, T)
    x, w = _rule(N, method)
    theta = np.pi / 2 * (x + 1)
    Elist = center + r * np.exp(1j * theta)
    dz = 1j * r * np.exp(1j * theta)
    weights = (np.pi / 2) * w * fermi(Elist, mu, T) * dz
    return Elist, weights


def broadening_grid(mu, N, T, method='ant'):
    """Extra real-axis segment [mu-10kT, mu+10kT] with N//8 points (density.py:730-742)."""
    broadening = 10 * kB * T
    Nbroad = int(N // 8)
    if method in ('legendre', 'chebyshev', 'ant'):
        x_fermi, w_fermi = roots_legendre(Nbroad)
    else:
        x_fermi = np.linspace(-1, 1, Nbroad)
        w_fermi = 2 * np.ones(Nbroad) / Nbroad
    Elist = broadening * (x_fermi) + mu
    weights = broadening * w_fermi * fermi(Elist, mu, T)
    return Elist, weights


# ------------------------------------------------- energy-dependent densities
def densityRealN(F, S, g, Emin, mu, N=100, T=TEMPERATURE, showText=True):
    """Equilibrium density from a real-axis Gauss-Legendre grid (density.py:385-436)."""
    Elist, weights = real_axis_grid(Emin, mu, N, T)
    if showText:
        print(f'Integrating {N} points along real axis...')
    defInt = GrInt(F, S, g, Elist, weights)
    if showText:
        print('Integration done!')
    return (-1 + 0j) * np.imag(defInt) / (np.pi)


def densityReal(F, S, g, Emin, mu, tol=ADAPTIVE_INTEGRATION_TOL, T=TEMPERATURE, maxN=MAX_CYCLES, debug=False):
    """Doubling wrapper around densityRealN (density.py:438-484)."""
    P = np.zeros_like(F)
    N = 1
    maxDP = 1e9
    while N < maxN:
        P_ = P.copy()
        P = densityRealN(F, S, g, Emin, mu, N, T, showText=False)
        maxDP = np.max(np.abs(P - P_))
        if maxDP < tol:
            print(f'Adaptive integration converged to {maxDP:.3e} in {N} points.')
            return P
        N *= 2
    print(f'Warning: adaptive integration not converged after {maxN} points: maxDP={maxDP:.2E}')
    return P


def densityGridN(F, S, g, mu1, mu2, ind=None, N=100, T=TEMPERATURE, showText=True):
    """Non-equilibrium (bias-window) density, Gauss-Legendre (density.py:487-544)."""
    energies, weights = bias_window_grid(mu1, mu2, N, T)
    if showText:
        print(f'Real integration over {N} points...')
    den = GrLessInt(F, S, g, energies, weights, ind)
    if showText:
        print('Integration done!')
    return den / (2 * np.pi)


def densityGridTrap(F, S, g, mu1, mu2, ind=None, N=100, T=TEMPERATURE):
    """Midpoint-on-a-uniform-grid variant (density.py:546-603): the reference loops over
    interval midpoints; the same midpoints and dFermi*dE*dInt weights go to GrLessInt."""
    muLo, muHi, dInt, Emin, Emax = _window(mu1, mu2, T)
    Egrid = np.linspace(Emin, Emax, N)
    print(f'Real integration over {N} points...')
    E = (Egrid[1:] + Egrid[:-1]) / 2
    dE = Egrid[1:] - Egrid[:-1]
    dFermi = fermi(E, muHi, T) - fermi(E, muLo, T)
    den = GrLessInt(F, S, g, E, dFermi * dE * dInt, ind)
    print('Integration done!')
    return den / (2 * np.pi)


def densityGrid(F, S, g, mu1, mu2, ind=None, tol=ADAPTIVE_INTEGRATION_TOL, T=TEMPERATURE, debug=False):
    """Adaptive (nested ANT) bias-window density (density.py:605-658)."""
    muLo, muHi, dInt, Emin, Emax = _window(mu1, mu2, T)
    mid = (Emax - Emin) / 2

    def computePoint(x, w):
        E = mid * (x + 1) + Emin
        dFermi = fermi(E, muHi, T) - fermi(E, muLo, T)
        weights = mid * w * dFermi * dInt
        return GrLessInt(F, S, g, E, weights, ind)

    den = integratePointsAdaptiveANT(computePoint, tol=tol, debug=debug)
    if debug:
        print('Integration done!')
    return den / (2 * np.pi)


def densityComplexN(F, S, g, Emin, mu, N=100, T=TEMPERATURE, showText=True, method='ant'):
    """Equilibrium density from the complex contour (density.py:660-748)."""
    Elist, weights = contour_grid(Emin, mu, N, T, method)
    if showText:
        print(f'Complex Integration over {N} points...')
    lineInt = GrInt(F, S, g, Elist, weights)
    if T > 0:
        if showText:
            print('Integrating Fermi Broadening')
        Eb, wb = broadening_grid(mu, N, T, method)
        lineInt += GrInt(F, S, g, Eb, wb)
    if showText:
        print('Integration done!')
    return (1 + 0j) * np.imag(lineInt) / np.pi


def densityComplex(F, S, g, Emin, mu, tol=ADAPTIVE_INTEGRATION_TOL, T=TEMPERATURE, debug=False):
    """Adaptive contour integration (density.py:750-816)."""
    broadening, center, r = _contour(Emin, mu, T)

    def computePoint(x, w):
        theta = np.pi / 2 * (x + 1)
        z = center + r * np.exp(1j * theta)
        dz = 1j * r * np.exp(1j * theta)
        weights = (np.pi / 2) * w * dz * fermi(z, mu, T)
        return GrInt(F, S, g, z, weights)

    print('Complex Contour Integration:')
    lineInt = integratePointsAdaptiveANT(computePoint, tol=tol, debug=debug)
    if T > 0:
        print('Integrating Fermi Broadening:')

        def computePointBroadening(x, w):
            E = broadening * (x) + mu
            weights = broadening * w * fermi(E, mu, T)
            return GrInt(F, S, g, E, weights)

        lineInt += integratePointsAdaptiveANT(computePointBroadening, tol=tol, debug=debug)
    return (1 + 0j) * np.imag(lineInt) / np.pi


# ------------------------------------------------------------- DOS at one E
def _compute_dos_at_energy(E, F, S, sigma_total):
    """-Im Tr G / pi at one energy with an explicit Sigma (density.py:49-54)."""
    from .engine import get_engine
    eng = get_engine()
    eng.set_system(F, S)
    h = eng.sigma_precomputed(np.asarray(sigma_total)[None])
    try:
        return float(eng.dos(h, [E], per_site=False)[0])
    finally:
        eng.sigma_free(h)
