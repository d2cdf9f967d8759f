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
import os
import numpy as np
from scipy.special import roots_legendre as _scipy_roots_legendre
import functools


@functools.lru_cache(maxsize=64)
def _legendre_nodes(N):
    x, w = _scipy_roots_legendre(N)
    x.setflags(write=False); w.setflags(write=False)
    return x, w


def roots_legendre(N):
    """scipy.special.roots_legendre, the nodes of each order kept (an SCF run asks for the same few orders in every
    density step: 9 calls, 2 ms of an N = 60 step); read-only arrays, the same bits."""
    return _legendre_nodes(int(N))

from .config import (TEMPERATURE, ADAPTIVE_INTEGRATION_TOL, N_KT, MAX_CYCLES, MAX_GRID_POINTS)
from . import integrate as _integrate
from .integrate import GrInt, GrLessInt, GrIntSegments, GrLessIntSegments

_ENGINE_GRLESSINT = GrLessInt
_ENGINE_GRINT = GrInt     # speculation over several levels only while GrInt is the engine's own: a rebound name
                          # (the bookkeeping spies, an oracle-served replay) sees the reference's call sequence

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


SPECULATIVE_POINTS = 64      # new nodes evaluated per launch AHEAD of the convergence test (0: level by level);
SPECULATIVE_POINTS_ONE_CU = int(os.environ.get("NEGF_SPECULATIVE_ONE_CU", "192"))  # ... for systems whose inverse is ONE workgroup per matrix (n <= 256): up to one matrix
                                 # per compute unit a launch takes what a single matrix does (n = 200: 2 points 0.61 ms,
                                 # 108 points 0.68 ms, 324 points 1.44 ms) -- levels 2 ... 162 in one go
SPECULATIVE_POINTS_SMALL = int(os.environ.get("NEGF_SPECULATIVE_SMALL", "512"))   # ... and for n <= 96 (matrix in registers, several per compute unit; n = 60: 2 points
                                 # 93 us, 324 points 132 us): every level of the rule, 486 points, in one go


def _speculation_budget(F, S=None, g=None):
    """How many new nodes an adaptive integration may evaluate ahead of its convergence test, for this system: zero
    (level by level, the reference's own sequence) unless the levels really are fused into one pass of the engine
    (integrate.can_fuse_segments: a device-lowerable ``g``; per rank and with one all-reduce under energy sharding, per
    block for a spin-block system) -- otherwise every speculated level would cost a launch of its own, up to 486 nodes
    where the reference stops after 18."""
    if SPECULATIVE_POINTS <= 0:
        return 0
    if g is not None and not _integrate.can_fuse_segments(F, S, g):
        return 0
    n = np.shape(F)[0]
    return SPECULATIVE_POINTS_SMALL if n <= 96 else SPECULATIVE_POINTS_ONE_CU if n <= 256 else SPECULATIVE_POINTS


_ANT_LEVELS = {}


def _ant_levels(maxN):
    """The levels N = 2, 6, 18, ... <= maxN of the nested rule as (N, new nodes, their weights, nested-weight ratio):
    the node bookkeeping of density.py:239-252 (old nodes recognised by value, rounded to 14 digits).  Depends on maxN
    only: built once per value (an SCF step runs ~50 adaptive integrations)."""
    if maxN in _ANT_LEVELS:
        return _ANT_LEVELS[maxN]
    out = []
    prev_x = prev_sumW = None
    N = 2
    while N <= maxN:
        x, w = getANTPoints(N)
        if prev_x is None:
            out.append((N, x[0:2], w[0:2], None))
        else:
            old_mask = np.isin(np.round(x, 14), np.round(prev_x, 14))
            assert int(old_mask.sum()) == prev_x.size, "Old nodes mismatch"
            ratio = float(np.sum(w[old_mask]) / prev_sumW)
            new_mask = ~old_mask
            out.append((N, x[new_mask], w[new_mask], ratio))
        prev_x = x
        prev_sumW = float(np.sum(w))
        N *= 3
    for _, xs, ws, _ in out:
        xs.setflags(write=False); ws.setflags(write=False)
    _ANT_LEVELS[maxN] = out
    return out


def _adaptive_ant_steps(levels, tol, budget, direct=None):
    """The refinement itself, as a generator: yields the list of level indices it wants evaluated next (the level it
    needs and, while they fit ``budget`` nodes, the ones after it), receives ``{index: value}``, and returns the integral
    -- the update and the stopping test of density.py:239-268, level by level, whoever evaluates the nodes and in
    whatever company.  ``direct(N)`` (debug only) evaluates the whole level-N rule for the reference's debug prints."""
    ahead = {}
    P = new_P = None
    bufs = mag = None
    N = 2
    maxDP = 1e10
    for i, (N, x_new, w_new, ratio) in enumerate(levels):
        if i not in ahead:
            group, pts = [], 0
            for j in range(i, len(levels)):
                if group and pts + levels[j][1].size > budget:
                    break
                group.append(j); pts += levels[j][1].size
            ahead.update((yield group))
        if ratio is None:
            P = ahead.pop(i)
        else:
            # (the same operations as new_P = P * ratio; new_P += value; max|new_P - P|, written into work arrays that
            #  live as long as this refinement: a fresh 10-MB array per operation and level is page faults, not arithmetic)
            inc = ahead.pop(i)
            if np.ndim(P) == 2 and np.shape(inc) == np.shape(P):
                if bufs is None:
                    dt = np.result_type(P, inc, type(ratio))
                    bufs = [np.empty(np.shape(P), dtype=dt) for _ in range(3)]
                    mag = np.empty(np.shape(P), dtype=np.abs(np.zeros(1, dtype=dt)).dtype)
                new_P = bufs[0] if P is not bufs[0] else bufs[1]
                np.multiply(P, ratio, out=new_P)
                new_P += inc
                np.subtract(new_P, P, out=bufs[2])
                np.abs(bufs[2], out=mag)
                maxDP = np.max(mag)
            else:
                new_P = P * ratio
                new_P += inc
                maxDP = np.max(np.abs(new_P - P))
            if direct is not None:
                full = direct(N)
                print(f"N={N}, nested-weight ratio ~ {ratio:.3f}, maxDP={maxDP:.3e}")
                print(f"Direct Calculation: N={N}, maxDP={np.max(np.abs(full - P)):.3e}, "
                      f"maxDiff={np.max(np.abs(full - new_P)):.3e}")
            P = new_P                                   # (a fresh array every level: P * ratio allocates; no copy needed)
            if maxDP < tol:
                print(f'Adaptive integration converged to {maxDP:.3e} in {N} points.')
                return new_P
    print(f'Adaptive integration reached full grid ({N / 1} points), final error {maxDP:.3e}')
    return new_P


def integratePointsAdaptiveANT(computePoint, tol=ADAPTIVE_INTEGRATION_TOL, maxN=MAX_GRID_POINTS, debug=False,
                               computeLevels=None, budget=None):
    """Adaptive nested quadrature (density.py:211-273): levels N = 2, 6, 18, ...; each
    level sends ONLY its new nodes to ``computePoint(x, w)`` and rescales the running
    value by the nested-weight ratio; stops when max|dP| < tol or N would exceed maxN.

    ``computeLevels([(x, w), ...]) -> [value, ...]`` (optional) evaluates the new nodes of SEVERAL levels in one go:
    the levels the refinement is about to visit are then requested together while they add up to at most
    ``budget`` nodes (default SPECULATIVE_POINTS: 2 + 4 + 12 + 36) -- on a GPU a launch of 2 ... 36 energy points costs
    what a launch of 54 does -- and consumed level by level with the reference's own update and stopping test; values
    computed past the level that converges are dropped."""
    levels = _ant_levels(maxN)
    if budget is None:
        budget = SPECULATIVE_POINTS
    if computeLevels is None or debug:
        budget = 0                                      # level by level: the reference's call sequence
    steps = _adaptive_ant_steps(levels, tol, budget,
                                (lambda N: computePoint(*getANTPoints(N))) if debug else None)
    try:
        want = next(steps)
        while True:
            if len(want) == 1:
                got = {want[0]: computePoint(levels[want[0]][1], levels[want[0]][2])}
            else:
                got = dict(zip(want, computeLevels([(levels[j][1], levels[j][2]) for j in want])))
            want = steps.send(got)
    except StopIteration as done:
        return done.value


def integrateJointlyAdaptiveANT(node_grids, computeSegments, tol=ADAPTIVE_INTEGRATION_TOL, maxN=MAX_GRID_POINTS, budget=None):
    """Several adaptive integrations of ONE system advanced together: ``node_grids[k](x, w) -> (energies, weights)`` maps
    the rule's nodes to integral k's grid, ``computeSegments([(E, w), ...]) -> [value, ...]`` evaluates any list of grids
    in one pass.  In every round each integration that is still refining asks for its next levels, all requests go out
    as ONE call, and each integration consumes its own values with its own stopping test -- the contour and the
    Fermi-tail integral of densityComplex share their launches that way.  Returns the list of integrals."""
    levels = _ant_levels(maxN)
    if budget is None:
        budget = SPECULATIVE_POINTS
    runs = [_adaptive_ant_steps(levels, tol, budget) for _ in node_grids]
    want, result = {}, [None] * len(runs)
    for k, r in enumerate(runs):
        want[k] = next(r)
    while want:
        owners = [(k, j) for k in sorted(want) for j in want[k]]
        values = computeSegments([node_grids[k](levels[j][1], levels[j][2]) for k, j in owners])
        got = {k: {} for k in want}
        for (k, j), v in zip(owners, values):
            got[k][j] = v
        for k in sorted(got):
            try:
                want[k] = runs[k].send(got[k])
            except StopIteration as done:
                result[k] = done.value
                del want[k]
    return result


REFINE_ON_DEVICE = os.environ.get("NEGF_REFINE_ON_DEVICE", "1") != "0"   # adaptive GrInt integrations: update + stopping test in the library
_CONVERGED_AT = {}          # (provider, tol) -> [level at which the arc / the Fermi tail of the last densityComplex converged]: hints only


_LEVEL_GROUPS = {}


def _level_group(maxN, lo, hi):
    """Levels lo ... hi-1 of the nested rule as ONE node array: (nodes, weights, nodes per level, ratio per level), read-only,
    built once.  A grid map applied to it evaluates all those levels in one go -- every map is elementwise, so each node gets
    the value it gets level by level (twelve small numpy pipelines per Fermi probe become two)."""
    key = (maxN, lo, hi)
    if key not in _LEVEL_GROUPS:
        lv = _ant_levels(maxN)[lo:hi]
        x = np.concatenate([l[1] for l in lv]); w = np.concatenate([l[2] for l in lv])
        x.setflags(write=False); w.setflags(write=False)
        _LEVEL_GROUPS[key] = (x, w, tuple(l[1].size for l in lv), tuple(l[3] for l in lv))
    return _LEVEL_GROUPS[key]


def _refine_jointly(node_grids, refine, tol=ADAPTIVE_INTEGRATION_TOL, maxN=MAX_GRID_POINTS, budget=None, hints=None):
    """integrateJointlyAdaptiveANT with the refinement itself on the device: every round hands the levels each integration
    is about to visit (up to ``budget`` new nodes, as _adaptive_ant_steps groups them) to ``refine`` -- ONE pass of the
    engine, which applies the reference's update and stopping test level by level (density.py:239-268) and returns the value
    at the level that converged, or after the last one together with the fact that it did not.  What the host no longer does:
    receive a sum per level and run five numpy passes over each.  ``refine`` takes [(E, w, nodes per level, ratio per level,
    running value or None)] (Engine.gr_int_refine).  Messages as in _adaptive_ant_steps.
    ``hints`` (optional, a list updated in place): per integration the level at which its predecessor converged -- the probes of
    a Fermi search are neighbours, an integration that converged at 162 nodes last time is not sent the 324 new nodes of the
    next level ahead of its test again (a third of the points of an N = 60 probe); if it does need them, it asks in a second
    round.  Only WHAT is evaluated ahead of the test changes, never the result."""
    levels = _ant_levels(maxN)
    if budget is None:
        budget = SPECULATIVE_POINTS
    K = len(node_grids)
    nxt, running, result = [0] * K, [None] * K, [None] * K
    active = list(range(K))
    while active:
        groups, requests = [], []
        for k in active:
            hi, pts = nxt[k], 0
            cap = len(levels) if hints is None or nxt[k] or hints[k] is None else min(len(levels), hints[k] + 1)
            while hi < cap and (hi == nxt[k] or pts + levels[hi][1].size <= budget):
                pts += levels[hi][1].size; hi += 1
            x, w, counts, ratios = _level_group(maxN, nxt[k], hi)
            groups.append((nxt[k], hi))
            requests.append(node_grids[k](x, w) + (counts, ratios, running[k]))
        still = []
        for k, (lo, hi), (value, conv, maxdps) in zip(active, groups, refine(requests, tol)):
            if conv >= 0:
                print(f'Adaptive integration converged to {maxdps[conv]:.3e} in {levels[lo + conv][0]} points.')
                result[k] = value
                if hints is not None:
                    hints[k] = lo + conv
            elif hi == len(levels):
                print(f'Adaptive integration reached full grid ({levels[-1][0] / 1} points), final error {maxdps[-1]:.3e}')
                result[k] = value
                if hints is not None:
                    hints[k] = len(levels) - 1
            else:
                running[k], nxt[k] = value, hi
                still.append(k)
        active = still
    return result


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
    """Doubling wrapper around densityRealN (density.py:438-484): N = 1, 2, 4, ... Gauss-Legendre points until two
    successive densities agree to tol.  The grids the doubling is about to visit are evaluated together while they add
    up to at most SPECULATIVE_POINTS points (1 + 2 + ... + 32: one launch instead of six) and compared in the
    reference's order; densities past the converged one are dropped."""
    P = np.zeros_like(F)
    N = 1
    maxDP = 1e9
    ahead = {}
    budget = _speculation_budget(F, S, g)
    while N < maxN:
        P_ = P.copy()
        if N not in ahead:
            group, pts, M = [], 0, N
            while M < maxN and (not group or pts + M <= budget):
                group.append(M); pts += M; M *= 2
            if len(group) == 1 or debug or GrInt is not _ENGINE_GRINT:
                group = [N]
                ahead[N] = densityRealN(F, S, g, Emin, mu, N, T, showText=False)
            else:
                sums = GrIntSegments(F, S, g, [real_axis_grid(Emin, mu, M, T) for M in group])
                for M, v in zip(group, sums):
                    ahead[M] = (-1 + 0j) * np.imag(v) / (np.pi)
        P = ahead.pop(N)
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

    def grid(x, w):
        E = mid * (x + 1) + Emin
        dFermi = fermi(E, muHi, T) - fermi(E, muLo, T)
        return E, mid * w * dFermi * dInt

    def computePoint(x, w):
        return GrLessInt(F, S, g, *grid(x, w), ind)

    def computeLevels(nodes):
        return GrLessIntSegments(F, S, g, [grid(x, w) for x, w in nodes], ind)

    den = integratePointsAdaptiveANT(computePoint, tol=tol, debug=debug,
                                     computeLevels=computeLevels if GrLessInt is _ENGINE_GRLESSINT else None,
                                     budget=_speculation_budget(F, S, g))
    if debug:
        print('Integration done!')
    return den / (2 * np.pi)


def densityComplexN(F, S, g, Emin, mu, N=100, T=TEMPERATURE, showText=True, method='ant'):
    """Equilibrium density from the complex contour (density.py:660-748).  At T > 0 the contour and the Fermi-broadening
    segment go to the engine as ONE pass (GrIntSegments) and are added in the reference's order."""
    Elist, weights = contour_grid(Emin, mu, N, T, method)
    if showText:
        print(f'Complex Integration over {N} points...')
    if T > 0 and GrInt is _ENGINE_GRINT:
        if showText:
            print('Integrating Fermi Broadening')
        lineInt, tail = GrIntSegments(F, S, g, [(Elist, weights), broadening_grid(mu, N, T, method)])
        lineInt = lineInt + tail
    else:
        lineInt = GrInt(F, S, g, Elist, weights)
        if T > 0:
            if showText:
                print('Integrating Fermi Broadening')
            Eb, wb = broadening_grid(mu, N, T, method)
            lineInt += GrInt(F, S, g, Eb, wb)
    if showText:
        print('Integration done!')
    return (1 + 0j) * np.imag(lineInt) / np.pi


def densityEquilibriumN(F, S, g, Eminf, Emin, mu, N_real=100, N_contour=100, T=TEMPERATURE, method='ant'):
    """(densityRealN(F, S, g, Eminf, Emin, N_real, T=0), densityComplexN(F, S, g, Emin, mu, N_contour, T)) -- the two
    fixed-grid integrals of a density step at a given Fermi level (scfE.py:316-328, :444-446) -- from ONE pass of the
    engine over the real-axis grid below Emin, the contour and its Fermi-broadening segment (GrIntSegments; sharded over
    the ranks of a multi-GPU run as one grid with one all-reduce)."""
    if GrInt is not _ENGINE_GRINT:
        return (densityRealN(F, S, g, Eminf, Emin, N_real, T=0, showText=False),
                densityComplexN(F, S, g, Emin, mu, N_contour, T, showText=False, method=method))
    segs = [real_axis_grid(Eminf, Emin, N_real, 0), contour_grid(Emin, mu, N_contour, T, method)]
    if T > 0:
        segs.append(broadening_grid(mu, N_contour, T, method))
    sums = GrIntSegments(F, S, g, segs)
    lineInt = sums[1] + sums[2] if T > 0 else sums[1]
    return (-1 + 0j) * np.imag(sums[0]) / (np.pi), (1 + 0j) * np.imag(lineInt) / np.pi


def densityComplex(F, S, g, Emin, mu, tol=ADAPTIVE_INTEGRATION_TOL, T=TEMPERATURE, debug=False):
    """Equilibrium density by ADAPTIVE integration (density.py:750-816): the nested ANT rule refined level by level
    (2, 6, 18, ... nodes; only the new nodes of a level are evaluated) on the upper half circle through Emin and
    mu - 10 kT, then -- at T > 0 -- on the real segment mu +- 10 kT that carries the Fermi tail.  Both pieces hand
    (energies, weights) to GrInt; node -> energy maps and weight products keep the reference's operation order (the
    captured grids are compared bit for bit, tests/test_grids.py)."""
    half_width, mid, radius = _contour(Emin, mu, T)
    quarter_turn = np.pi / 2
    budget = _speculation_budget(F, S, g) if GrInt is _ENGINE_GRINT else 0      # (once per call: it checks the system's structure)

    def on_arc(x, w):
        phase = np.exp(1j * (quarter_turn * (x + 1)))
        return mid + radius * phase, quarter_turn * w * (1j * radius * phase)

    def on_tail(x, w):
        return half_width * (x) + mu, half_width * w

    def grid_of(node_map):
        def grid(x, w):
            z, wz = node_map(x, w)
            return z, wz * fermi(z, mu, T)
        return grid

    def integral_over(node_map):
        grid = grid_of(node_map)
        return integratePointsAdaptiveANT(lambda x, w: GrInt(F, S, g, *grid(x, w)), tol=tol, debug=debug,
                                          computeLevels=(lambda nodes: GrIntSegments(F, S, g, [grid(x, w) for x, w in nodes]))
                                          if GrInt is _ENGINE_GRINT else None,
                                          budget=budget)

    fused = GrInt is _ENGINE_GRINT and not debug and budget > 0
    refine = _integrate.GrIntRefiner(F, S, g) if fused and REFINE_ON_DEVICE else None
    if T > 0 and fused:
        # the contour and the Fermi tail refine together: one launch per round for both (integrateJointlyAdaptiveANT; where
        # the library can run the refinement itself, _refine_jointly: only the refined values come back)
        print('Complex Contour Integration (with the Fermi broadening):')
        if refine is not None:
            hints = _CONVERGED_AT.setdefault((id(g), float(tol)), [None, None])
            if len(_CONVERGED_AT) > 64:
                _CONVERGED_AT.clear()
            total, tail = _refine_jointly([grid_of(on_arc), grid_of(on_tail)], refine, tol=tol, budget=budget, hints=hints)
            total = total + tail
        else:
            total, tail = integrateJointlyAdaptiveANT([grid_of(on_arc), grid_of(on_tail)],
                                                      lambda segs: GrIntSegments(F, S, g, segs), tol=tol,
                                                      budget=budget)
            total += tail
        return (1 + 0j) * np.imag(total) / np.pi
    print('Complex Contour Integration:')
    if refine is not None:
        total = _refine_jointly([grid_of(on_arc)], refine, tol=tol, budget=budget)[0]
    else:
        total = integral_over(on_arc)
    if T > 0:
        print('Integrating Fermi Broadening:')
        total += integral_over(on_tail)
    return (1 + 0j) * np.imag(total) / np.pi


# ------------------------------------------------------------- DOS at one E
# --------------------------------------------------------------------------- #
# closed-form density for energy-independent self-energies (density.py:276-382)
# --------------------------------------------------------------------------- #
def density(V, Vc, D, Gam, Emin, mu):
    """int_{Emin}^{mu} G Gamma G^H dE / 2 pi in closed form (Eq. 27 of PRB 65, 165401) for a constant
    self-energy: with  X (F + Sigma) X = V diag(D) V^-1  and  Vc = (V^H)^-1,
        P = V [ L o (Vc^H Gam Vc) ] V^H,   L_ij = (l_i - conj(l_j)) / (2 pi (D_i - conj(D_j))),
        l_i = log(1 - mu / D_i) - log(1 - Emin / D_i)            (complex logarithms),
    in the orthogonalised basis (density.py:276-329).  O(N^3) once, no energy grid: host numpy."""
    D = np.asarray(D)
    l = np.emath.log(1 - (mu / D)) - np.emath.log(1 - (Emin / D))
    L = (l[:, None] - l.conj()[None, :]) / (2 * np.pi * (D[:, None] - D.conj()[None, :]))
    return V @ (L * (Vc.conj().T @ Gam @ Vc)) @ V.conj().T


def bisectFermi(V, Vc, D, Gam, Nexp, conv=None, Eminf=None):
    """Bisection on the closed-form electron count between the lowest and highest level
    (density.py:331-382); prints the reference's messages."""
    from .config import FERMI_CALCULATION_TOL, ENERGY_MIN
    conv = FERMI_CALCULATION_TOL if conv is None else conv
    Eminf = ENERGY_MIN if Eminf is None else Eminf
    Emin, Emax = min(np.real(D)), max(np.real(D))
    dN, Niter, fermi_ = Nexp, 0, None
    while abs(dN) > conv and Niter < 1000:
        fermi_ = (Emin + Emax) / 2
        dN = np.trace(density(V, Vc, D, Gam, Eminf, fermi_)).real - Nexp
        if dN > 0:
            Emax = fermi_
        else:
            Emin = fermi_
        Niter += 1
    if Niter >= 1000:
        print('Warning: Bisection search timed out after 1000 iterations!')
    print(f'Bisection fermi search converged to {dN:.2E} in {Niter} iterations.')
    return fermi_


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


# --------------------------------------------------------------------------- #
# Integration limits and Fermi-level searches (gauNEGF/density.py:821-1515).
# Pure callers of the grid integrals above: every density evaluation inside the
# loops below is one GPU integral; F and S stay resident on the device between
# calls (Engine.set_system skips identical uploads), only mu changes.
# --------------------------------------------------------------------------- #
from .config import FERMI_CALCULATION_TOL, FERMI_SEARCH_CYCLES, ENERGY_MIN   # noqa: E402
from ._hostblas import limited_call                                          # noqa: E402

FERMI_DEBUG = False


def _orbital_energies(F, S, hermitian=False):
    """Sorted real parts of eig(inv(S) F) (density.py:822, 994-996, 1048)."""
    M = np.linalg.solve(np.asarray(S), np.asarray(F))
    vals = np.linalg.eigvalsh(M) if hermitian else np.linalg.eigvals(M)
    return np.sort(np.real(vals))


CALC_EMIN_ROUTE = None      # None: the environment (NEGF_CALC_EMIN_ROUTE) decides; "reference" | "fast"


def _calc_emin_route():
    r = CALC_EMIN_ROUTE if CALC_EMIN_ROUTE is not None else os.environ.get("NEGF_CALC_EMIN_ROUTE", "fast")
    if r not in ("fast", "reference"):
        raise ValueError(f"NEGF_CALC_EMIN_ROUTE / density.CALC_EMIN_ROUTE must be 'fast' or 'reference', not {r!r}")
    return r


def _lowest_orbital_energy(F, S):
    """min Re eig(inv(S) F) (density.py:822).  The reference takes it from the full non-symmetric eigenproblem; for a
    Hermitian F and a Hermitian positive-definite S of 256 orbitals or more the lowest GENERALISED eigenvalue is the
    same number to rounding (relative 1e-13, measured) and costs a tenth (N = 800: 17 ms against 153 ms, 12 % of a
    density step).  Smaller systems and anything not exactly Hermitian keep the reference's route, bit for bit.
    This is a DEVIATION in the last bits of Emin, and with it of every grid derived from Emin:
    ``NEGF_CALC_EMIN_ROUTE=reference`` (or ``density.CALC_EMIN_ROUTE = "reference"``) keeps the reference's route for
    every system, bit for bit (tests/test_adaptive_host.py::test_calc_emin_reference_route_is_bit_identical)."""
    F = np.asarray(F); S = np.asarray(S)
    if _calc_emin_route() == "fast" and F.shape[0] >= 256 and np.array_equal(F, F.conj().T) and np.array_equal(S, S.conj().T):
        try:
            from scipy.linalg import eigh
            return float(eigh(F, S, eigvals_only=True, subset_by_index=[0, 0], check_finite=False)[0])
        except (np.linalg.LinAlgError, ValueError):             # S not positive definite, non-finite input: the reference's route
            pass
    return min(_orbital_energies(F, S))


@limited_call
def calcEmin(F, S, g, tol=FERMI_CALCULATION_TOL, maxN=MAX_CYCLES):
    """Lower contour bound: walk down in 1 eV steps from (lowest orbital - 5 eV) until the
    DOS falls below tol (density.py:821-836).

    A provider that lives on the device serves the walk in batches: the energies the loop WOULD visit (the same repeated
    ``Emin -= 1``, so the same floating-point values) are evaluated 8, 16, 32, ... at a time by one DOS launch each and
    the first one at or below ``tol`` is taken -- same Emin, same sample count, without one provider upload, one
    single-matrix launch and one download per step (a 200-orbital matrix alone in a launch costs 0.7 ms of latency)."""
    Emin = _lowest_orbital_energy(F, S) - 5
    counter = 0
    if hasattr(g, "_negf_lower"):
        from .engine import get_engine
        eng = get_engine()
        eng.set_system(F, S)
        handle = g._negf_lower(eng)
        walk = [Emin]
        dP, chunk, done = None, 8, False
        while not done:
            while len(walk) < min(counter + chunk, maxN + 1):
                walk.append(walk[-1] - 1)
            vals = eng.dos(handle, np.array(walk[counter:counter + chunk]), per_site=False)
            for v in vals:
                dP = float(v)
                if not (dP > tol and counter < maxN):
                    done = True
                    break
                counter += 1
            else:
                done = counter >= len(walk) and len(walk) > maxN
            chunk *= 2
        Emin = walk[min(counter, len(walk) - 1)]
    else:
        dP = _compute_dos_at_energy(Emin, F, S, g.sigmaTot(Emin))
        while dP > tol and counter < maxN:
            Emin -= 1
            dP = _compute_dos_at_energy(Emin, F, S, g.sigmaTot(Emin))
            counter += 1
    if counter == maxN:
        print(f'Warning: Emin still not within tolerance (final value = {dP}) after {maxN} energy samples')
    print(f'Calculated Emin: {Emin} eV, DOS = {dP:.2E}')
    return Emin


def _double_until_converged(evaluate, start, tol, maxN, label):
    """Shared doubling loop of integralFit / integralFitNEGF: N <- 2N until the diagonal of
    the density stops changing by more than tol; returns the last N that was needed."""
    N = start
    dP = np.inf
    rho = None
    while dP > tol and N < maxN:
        N *= 2
        rho_ = np.real(evaluate(N))
        dP = max(abs(np.diag(rho_ - rho))) if rho is not None else max(abs(np.diag(rho_)))
        print(f"MaxDP = {dP:.2E}")
        rho = rho_
    if dP < tol:
        N /= 2
    elif N >= maxN and dP > tol:
        print(f'Warning: {label} still not within tolerance (final value = {dP})')
    print(f'Final {label}: {N}')
    return N


def integralFit(F, S, g, mu, Eminf=ENERGY_MIN, tol=FERMI_CALCULATION_TOL, T=TEMPERATURE, maxN=MAX_CYCLES):
    """(Emin, N1, N2): contour lower bound, contour points, real-axis points (density.py:838-914)."""
    Emin = calcEmin(F, S, g, tol, maxN)
    Ncomplex = _double_until_converged(
        lambda N: densityComplexN(F, S, g, Emin, mu, N, T=T, showText=False), 4, tol, maxN, 'Ncomplex')
    Nreal = _double_until_converged(
        lambda N: densityRealN(F, S, g, Eminf, Emin, N, T=0, showText=False), 8, tol, maxN, 'Nreal')
    return Emin, Ncomplex, Nreal


def integralFitNEGF(F, S, g, fermi, qV, Eminf=ENERGY_MIN, tol=FERMI_CALCULATION_TOL, T=TEMPERATURE,
                    maxGrid=MAX_GRID_POINTS):
    """Grid size of the bias-window integral (density.py:916-966)."""
    def both(N):
        rho = np.real(densityGridN(F, S, g, fermi, fermi + (qV / 2), ind=0, N=N, T=T, showText=False))
        return rho + np.real(densityGridN(F, S, g, fermi, fermi - (qV / 2), ind=-1, N=N, T=T, showText=False))
    return _double_until_converged(both, 8, tol, maxGrid, 'Nnegf')


def _count(P, S, nOrbs=0):
    PS = P @ S
    return np.trace(PS) if nOrbs == 0 else np.trace(PS[-nOrbs:, -nOrbs:])


def calcFermi(g, ne, Emin, Emax, fermiGuess=0, N1=100, N2=50, Eminf=ENERGY_MIN, T=TEMPERATURE,
              tol=FERMI_CALCULATION_TOL, maxcycles=MAX_CYCLES, nOrbs=0):
    """Bisection on the electron count of a contact (density.py:1054-1143).
    Returns (fermi, Emin, N1, N2)."""
    dos_eminf = _compute_dos_at_energy(Eminf, g.F, g.S, g.sigmaTot(Eminf))
    print(f'Eminf DOS = {dos_eminf}')
    fermi = fermiGuess

    def low(temp):
        if N2 is None:
            return densityReal(g.F, g.S, g, Eminf, Emin, tol, temp)
        return densityRealN(g.F, g.S, g, Eminf, Emin, int(N2), temp, showText=False)

    def upper(E):
        if N1 is None:
            return densityComplex(g.F, g.S, g, Emin, E, tol, T)
        return densityComplexN(g.F, g.S, g, Emin, E, int(N1), T, showText=False, method='legendre')

    nELow = _count(low(T), g.S, nOrbs)
    print(f'Electrons below lowest onsite energy: {nELow}')
    if nELow >= ne:
        raise Exception('Calculated Fermi energy is below lowest orbital energy!')
    Ncurr = -1
    counter = 0
    lBound, uBound = Emin, Emax
    print('[getFermiContact] bisecting for the contact Fermi level')
    while abs(ne - Ncurr) > tol and uBound - lBound > tol / 10 and counter < maxcycles:
        g.setF(g.F, fermi, fermi)
        p_ = np.real(low(0) + upper(fermi))
        Ncurr = _count(p_, g.S, nOrbs)
        dN = ne - Ncurr
        if dN > 0 and fermi > lBound:
            lBound = fermi
        elif dN < 0 and fermi < uBound:
            uBound = fermi
        if abs(ne - Ncurr) > tol:
            fermi = (uBound + lBound) / 2
        print("DN:", dN, "Fermi:", fermi, "Bounds:", lBound, uBound)
        counter += 1
    if abs(ne - Ncurr) > tol and counter > maxcycles:
        print(f'[getFermiContact] tolerance not reached: Ef = {fermi:.2f} eV with N = {Ncurr:.2f}')
    print(f'[getFermiContact] {counter} bisection steps, Ef = {fermi:.2f} eV')
    return fermi, Emin, N1, N2


def getFermiContact(g, ne, tol=FERMI_CALCULATION_TOL, Eminf=ENERGY_MIN, maxcycles=MAX_CYCLES, T=TEMPERATURE,
                    nOrbs=0):
    """Fermi level of a contact (Bethe cluster or chain) from its electron count
    (density.py:969-1003)."""
    orbs = _orbital_energies(g.F, g.S)
    fermi = (orbs[int(ne) - 1] + orbs[int(ne)]) / 2
    Emin, N1, N2 = integralFit(g.F, g.S, g, fermi, Eminf, tol, T, maxN=maxcycles)
    return calcFermi(g, ne, Emin, max(orbs), fermi, N1, N2, Eminf, T, tol, maxcycles, nOrbs)[0]


def getFermi1DContact(gSys, ne, ind=0, tol=FERMI_CALCULATION_TOL, Eminf=ENERGY_MIN, T=TEMPERATURE,
                      maxcycles=MAX_CYCLES):
    """Fermi level of a 1-D chain lead: build the periodic lead from contact ``ind`` of gSys and
    search its Fermi level (density.py:1005-1052).  Returns (fermi, Emin, N1, N2)."""
    from .surfG1D import surfG
    F = np.asarray(gSys.aList[ind]); S = np.asarray(gSys.aSList[ind])
    tau = np.asarray(gSys.bList[ind]); stau = np.asarray(gSys.bSList[ind])
    inds = np.arange(len(F))
    g = surfG(F, S, [inds], [tau], [stau], eta=1e-6)
    Forbs = np.block([[F, tau], [tau.conj().T, F]])
    Sorbs = np.block([[S, stau], [stau.T, S]])
    gorbs = surfG(Forbs, Sorbs, [inds], [tau], [stau], eta=1e-6)
    orbs = _orbital_energies(Forbs, Sorbs, hermitian=False)
    fermi = (orbs[2 * int(ne) - 1] + orbs[2 * int(ne)]) / 2
    Emin, N1, N2 = integralFit(Forbs, Sorbs, gorbs, fermi, Eminf, tol, T, maxN=maxcycles)
    return calcFermi(g, ne, Emin, max(orbs), fermi, N1, N2, Eminf, T, tol, maxcycles)


def _mu_density(g, Emin, N, tol, T):
    """P(mu) evaluator shared by the searches below (density.py:1150-1153 and alike)."""
    if N is None:
        return lambda E: densityComplex(g.F, g.S, g, Emin, E, tol, T)
    return lambda E: densityComplexN(g.F, g.S, g, Emin, E, int(N), T, showText=False)


def calcFermiBisect(g, ne, Emin, Ef, N, tol=ADAPTIVE_INTEGRATION_TOL, conv=FERMI_CALCULATION_TOL,
                    maxcycles=FERMI_SEARCH_CYCLES, T=TEMPERATURE, uBound=None, lBound=None):
    """Bracket with DOS-sized steps, then bisect (density.py:1145-1207).  Returns (Ef, dE, P).
    (The reference passes (S, F) swapped to its DOS kernel at :1176; the intended (F, S) is used.)"""
    assert ne < len(g.F), "Number of electrons cannot exceed number of basis functions!"
    pMu = _mu_density(g, Emin, N, tol, T)
    E = Ef
    dE = tol
    counter = 0
    g.setF(g.F, E, E)
    P = pMu(E)
    Ncurr = np.trace(P @ g.S).real
    while None in [uBound, lBound] and counter < maxcycles:
        if Ncurr > ne:
            uBound = E
            Ef = uBound
            E -= dE
        if Ncurr < ne:
            lBound = E
            Ef = lBound
            E += dE
        dos = _compute_dos_at_energy(E, g.F, g.S, g.sigmaTot(E))
        dE = max(2 * abs(Ncurr - ne) / dos, dE)
        counter += 1
        g.setF(g.F, E, E)
        P = pMu(E)
        Ncurr = np.trace(P @ g.S).real
    while abs(ne - Ncurr) > conv and counter < maxcycles and uBound != lBound:
        dN = ne - Ncurr
        if dN > 0 and Ef > lBound:
            lBound = Ef
        elif dN < 0 and Ef < uBound:
            uBound = Ef
        Ef = (uBound + lBound) / 2
        dE = uBound - lBound
        counter += 1
        if abs(dN) > conv:
            g.setF(g.F, Ef, Ef)
            P = pMu(Ef)
            Ncurr = np.trace(P @ g.S)
    if counter == maxcycles:
        print(f'[calcFermiBisect] stopped at the cycle limit; |N - ne| = {abs(Ncurr-ne):.2E}')
    elif uBound == lBound:
        print(f'[calcFermiBisect] the bracket closed without reaching the tolerance; |N - ne| = {abs(Ncurr-ne):.2E}')
    return Ef, dE, P


def calcFermiSecant(g, ne, Emin, Ef, N, tol=ADAPTIVE_INTEGRATION_TOL, conv=FERMI_CALCULATION_TOL,
                    maxcycles=FERMI_SEARCH_CYCLES, T=TEMPERATURE):
    """Secant iteration on N(mu) - ne (density.py:1209-1247).  Returns (Ef, dE, P, |dN|)."""
    assert ne < len(g.F), "Number of electrons cannot exceed number of basis functions!"
    pMu = _mu_density(g, Emin, N, tol, T)
    g.setF(g.F, Ef, Ef)
    P = pMu(Ef)
    nCurr = np.trace(P @ g.S).real
    dE = conv
    counter = 0
    while abs(nCurr - ne) > conv and counter < maxcycles:
        Ef += dE
        g.setF(g.F, Ef, Ef)
        P = pMu(Ef)
        nNext = np.trace(P @ g.S).real
        if abs(nNext - nCurr) < 1e-10:
            print('[calcFermiSecant] the electron count did not respond to the step: trying a tenth of it')
            dE *= 0.1
            counter += 1
            continue
        dE = dE * ((ne - nCurr) / (nNext - nCurr)) - dE
        nCurr = nNext
        counter += 1
    Ef += dE
    if counter == maxcycles:
        print(f'[calcFermiSecant] stopped at the cycle limit; |N - ne| = {abs(nCurr-ne):.2E}')
    return Ef, dE, P, abs(nCurr - ne)


def _track_bounds(n, E, uBound, lBound):
    if n > 0:
        uBound = min(uBound, E) if uBound is not None else E
    elif n < 0:
        lBound = max(lBound, E) if lBound is not None else E
    return uBound, lBound


def calcFermiMuller(g, ne, Emin, Ef, N, tol=ADAPTIVE_INTEGRATION_TOL, conv=FERMI_CALCULATION_TOL,
                    maxcycles=FERMI_SEARCH_CYCLES, T=TEMPERATURE):
    """Muller's method from three points Ef, Ef -+ conv (density.py:1249-1330).
    Returns (E, dE, P, |dN|, uBound, lBound)."""
    assert ne < len(g.F), "Number of electrons cannot exceed number of basis functions!"
    pMu = _mu_density(g, Emin, N, tol, T)
    E2, E1, E0 = Ef, Ef - conv, Ef + conv
    uBound = lBound = None
    nList = []
    P = None
    for E in [E2, E1, E0]:
        g.setF(g.F, E, E)
        P = pMu(E)
        n = np.trace(P @ g.S).real - ne
        uBound, lBound = _track_bounds(n, E, uBound, lBound)
        if abs(n) < conv:
            return E, 0, P, abs(n), uBound, lBound
        nList.append(n)
    n2, n1, n0 = nList
    counter = 3
    dE = 0.0
    while counter < maxcycles:
        h0, h1 = E0 - E2, E1 - E2
        c = n2
        e0, e1 = n0 - c, n1 - c
        det = h0 * h1 * (h0 - h1)
        a = (e0 * h1 - h0 * e1) / det
        b = (h0 * h0 * e1 - h1 * h1 * e0) / det
        disc = np.sqrt(b * b - 4 * a * c) if b * b > 4 * a * c else 0
        if b < 0:
            disc = -disc
        dE = -2 * c / (b + disc)
        Enext = E2 + dE
        if abs(Enext - E1) < abs(Enext - E0):
            E0, E1 = E1, E0
            n0, n1 = n1, n0
        if abs(Enext - E2) < abs(Enext - E1):
            E1, n1 = E2, n2
        E2 = Enext
        g.setF(g.F, E2, E2)
        P = pMu(E2)
        n2 = np.trace(P @ g.S).real - ne
        uBound, lBound = _track_bounds(n2, E2, uBound, lBound)
        if abs(n2) < conv:
            break
        counter += 1
    if counter == maxcycles:
        print(f'[calcFermiMuller] stopped at the cycle limit; |N - ne| = {abs(n2):.2E}')
    return E2, dE, P, abs(n2), uBound, lBound


def calcFermiPolyFit(g, ne, Emin, Ef, N, tol=ADAPTIVE_INTEGRATION_TOL, conv=FERMI_CALCULATION_TOL,
                     maxcycles=FERMI_SEARCH_CYCLES, T=TEMPERATURE, order=3):
    """Accumulate (E, N-ne) points, fit a robust (Huber) polynomial through a PCHIP-smoothed
    version of them and step to its root nearest the last point, enforcing monotonicity
    (density.py:1332-1515).  Returns (E, dE, P, |dN|, uBound, lBound)."""
    from scipy.optimize import least_squares
    from scipy.interpolate import PchipInterpolator
    assert ne < len(g.F), "Number of electrons cannot exceed number of basis functions!"
    pMu = _mu_density(g, Emin, N, tol, T)

    def evaluate(E):
        g.setF(g.F, E, E)
        P = pMu(E)
        return P, np.trace(P @ g.S).real - ne

    E_pts, n_pts = [], []
    uBound = lBound = None
    E = Ef
    P, n = evaluate(E)
    if abs(n) < conv:
        return E, 0, P, abs(n), uBound, lBound
    E_pts.append(E); n_pts.append(n)
    step = conv * 10
    n_first = n
    counter = 1
    while counter < maxcycles:              # a second point with a resolvable change of N
        E = Ef + step
        P, n = evaluate(E)
        uBound, lBound = _track_bounds(n, E, uBound, lBound)
        if abs(n) < conv:
            return E, step, P, abs(n), uBound, lBound
        if n - n_first > 0:
            break
        step *= 10
        counter += 1
    E_pts.append(E); n_pts.append(n)
    dE = step
    while counter < maxcycles:
        poly_order = min(len(n_pts) - 1, order)
        Esort, nsort = list(zip(*sorted(zip(E_pts, n_pts))))
        n_smooth = PchipInterpolator(Esort, nsort)(E_pts)
        p0 = np.polyfit(E_pts, n_pts, poly_order)
        fit = least_squares(lambda cf: np.polyval(cf, E_pts) - n_smooth, p0, loss='huber',
                            f_scale=ADAPTIVE_INTEGRATION_TOL)
        roots = np.roots(fit.x)
        E_next = roots[np.argmin(np.abs(roots - E_pts[-1]))].real
        if (n_pts[-1] > 0 and E_next > E_pts[-1]) or (n_pts[-1] < 0 and E_next < E_pts[-1]):
            # the fit violated "higher E -> higher N": drop the last point, step the right way
            E_next = E_pts[-1] - np.sign(n_pts[-1]) * abs(dE) * 10
            E_pts.pop(); n_pts.pop()
            counter -= 1
        E = E_next
        P, n = evaluate(E)
        uBound, lBound = _track_bounds(n, E, uBound, lBound)
        E_pts.append(E); n_pts.append(n)
        dE = E - E_pts[-2]
        if abs(n) < conv:
            break
        counter += 1
    if counter >= maxcycles:
        print(f'[calcFermiPolyFit] stopped at the cycle limit; |N - ne| = {abs(n):.2E}')
    return E, dE, P, abs(n), uBound, lBound
