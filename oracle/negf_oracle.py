"""
numpy CPU oracle for the NEGF energy-grid hot path.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  Every function cites the reference lines it restates
(paths relative to the reference checkout, wliverno/GauNEGF @ 2025-11-21).

Plain loops over energy, ``np.linalg.solve(A, I)`` for every inverse (that is what
``gauNEGF/utils.py:52-54`` defines as ``inv``) and ``@`` for every product.
"""
import numpy as np

__all__ = [
    "inv", "gr_point", "gless_point", "GrInt", "GrLessInt", "gr_batch",
    "transmission_restricted", "transmission_spin_block", "dos_kernel",
    "dos_at_energy", "form_sigma", "chain1d_g", "chain1d_sigma_block",
    "chain1d_sigma", "chain1d_sigma_total", "bethe_sigmaK", "bethe_sigma_surface",
    "bethe_atom_sigma", "bethe_contact_sigma", "bethe_cluster_sigma_total",
    "fermi", "ant_points", "real_axis_grid", "bias_window_grid", "contour_grid",
    "broadening_grid", "adaptive_ant", "current_grid", "current_from_transmission",
    "ConstSigma", "Chain1DSigma", "kB", "eoverh", "N_KT", "DIM", "density_analytic",
    "density_analytic_from_system",
]

kB = 8.617e-5          # eV/K      (density.py:61, transport.py:36)
eoverh = 3.874e-5      # A/eV      (transport.py:35)
N_KT = 10              # config.py:20
DIM = 9                # surfGBethe.py:42


# --------------------------------------------------------------------------- #
# dense per-energy kernels
# --------------------------------------------------------------------------- #
def inv(A):
    """utils.py:52-54 -- the reference's inverse is solve(A, I), not LAPACK getri."""
    A = np.asarray(A)
    return np.linalg.solve(A, np.eye(A.shape[0]))


def gr_point(sigTot, E, F, S):
    """integrate.py:67-71 -- G^r(E) = solve(E S - F - Sigma_tot, I)."""
    return inv(E * S - F - sigTot)


def gless_point(sig, sigTot, E, F, S):
    """integrate.py:74-82 -- G (i (sig - sig^H)) G^H with G from gr_point."""
    G = gr_point(sigTot, E, F, S)
    gamma = 1j * (sig - np.conj(sig).T)
    return G @ gamma @ np.conj(G).T


def gr_batch(F, S, g, Elist):
    """[M,N,N] stack of G^r(E_m); for per-energy parity checks only."""
    F = np.asarray(F); S = np.asarray(S)
    return np.stack([gr_point(np.asarray(g.sigmaTot(E)), E, F, S) for E in Elist])


def GrInt(F, S, g, Elist, weights):
    """integrate.py:146-173 (+ _GInt :84-142) -- sum_m w_m G^r(E_m).

    Same serial form as the reference's own check,
    tests/test_computation_consistency.py:80-90."""
    Elist = np.asarray(Elist); weights = np.asarray(weights)
    F = np.asarray(F); S = np.asarray(S)
    assert Elist.size == weights.size, "Elist and weights must have the same length"
    assert F.shape == S.shape, "F and S must have the same shape"
    assert F.shape[0] == F.shape[1], "F and S must be square matrices"
    acc = np.zeros(F.shape, dtype=complex)
    for E, w in zip(Elist, weights):
        acc += w * gr_point(np.asarray(g.sigmaTot(E)), E, F, S)
    return acc


def GrLessInt(F, S, g, Elist, weights, ind=None):
    """integrate.py:177-208 -- sum_m w_m G Gamma_c G^H; Sigma_c = Sigma_tot when
    ind is None else g.sigma(E, ind).  Serial form as
    tests/test_computation_consistency.py:92-118."""
    Elist = np.asarray(Elist); weights = np.asarray(weights)
    F = np.asarray(F); S = np.asarray(S)
    assert Elist.size == weights.size, "Elist and weights must have the same length"
    assert F.shape == S.shape, "F and S must have the same shape"
    assert F.shape[0] == F.shape[1], "F and S must be square matrices"
    acc = np.zeros(F.shape, dtype=complex)
    for E, w in zip(Elist, weights):
        sigTot = np.asarray(g.sigmaTot(E))
        sig = sigTot if ind is None else np.asarray(g.sigma(E, ind))
        acc += w * gless_point(sig, sigTot, E, F, S)
    return acc


def transmission_restricted(E, F, S, sigma_total, gamma1, gamma2):
    """transport.py:150-157 -- Re Tr[(G1 G G2) G^H]."""
    G = inv(E * S - F - sigma_total)
    tmp = gamma1 @ G @ gamma2
    return float(np.real(np.trace(tmp @ np.conj(G).T)))


def transmission_spin_block(E, F, S, sigma_total, gamma1, gamma2):
    """transport.py:159-181 -- four spin-block traces of a 2N x 2N G.

    Blocks [uu, ud, du, dd] of G AND of G^H are sliced at the same positions
    (so 'ud' pairs G[:N,N:] with (G^H)[:N,N:]); gamma1 blocks [uu,uu,dd,dd],
    gamma2 blocks [uu,dd,uu,dd]."""
    G = inv(E * S - F - sigma_total)
    Ga = np.conj(G).T
    N = F.shape[0] // 2
    sl = [(slice(0, N), slice(0, N)), (slice(0, N), slice(N, None)),
          (slice(N, None), slice(0, N)), (slice(N, None), slice(N, None))]
    g1 = [gamma1[:N, :N], gamma1[:N, :N], gamma1[N:, N:], gamma1[N:, N:]]
    g2 = [gamma2[:N, :N], gamma2[N:, N:], gamma2[:N, :N], gamma2[N:, N:]]
    T = np.array([np.real(np.trace(g1[k] @ G[sl[k]] @ g2[k] @ Ga[sl[k]])) for k in range(4)])
    return float(np.sum(T)), T


def dos_kernel(E, F, S, sigma_total):
    """transport.py:183-190 -- (-Im diag G / pi) and its sum (no S weighting)."""
    G = inv(E * S - F - sigma_total)
    per_site = -np.imag(np.diag(G)) / np.pi
    return float(np.sum(per_site)), per_site


def dos_at_energy(E, F, S, sigma_total):
    """density.py:49-54 -- -Im Tr G / pi."""
    G = inv(E * S - F - sigma_total)
    return float(-np.imag(np.trace(G)) / np.pi)


# --------------------------------------------------------------------------- #
# constant self-energy (surfGTester.py:62-132, matTools.py:39-74)
# --------------------------------------------------------------------------- #
def form_sigma(inds, V, nsto, S=0):
    """matTools.py:39-74 -- -i 1e-9 S everywhere, then V on the contact
    diagonal (scalar V) or on the ix_(inds, inds) block (matrix V)."""
    if isinstance(S, int):
        S = np.eye(nsto)
    sigma = np.array(-1j * 1e-9 * S, dtype=complex)
    if isinstance(V, (int, complex, float)):
        for i in inds:
            sigma[i, i] = V
    else:
        sigma[np.ix_(inds, inds)] = V
    return sigma


class ConstSigma:
    """Energy-independent provider with the surfGTest interface
    (surfGTester.py:94-132): sigma(E,i) = sig[i]; sigmaTot = sum_i sigma(E,i)."""
    def __init__(self, F, S, indsList, sig1, sig2=None):
        self.F = F; self.S = S; self.N = len(F); self.indsList = indsList
        s2 = sig1 if sig2 is None else sig2
        self.sig = [form_sigma(indsList[0], sig1, self.N, S),
                    form_sigma(indsList[1], s2, self.N, S)]

    def sigma(self, E, i, conv=None):
        return self.sig[i]

    def sigmaTot(self, E, conv=None):
        tot = np.zeros((self.N, self.N), dtype=complex)
        for i in range(len(self.indsList)):
            tot += self.sigma(E, i)
        return tot

    def setF(self, F, mu1=None, mu2=None):
        self.F = F


# --------------------------------------------------------------------------- #
# 1-D chain decimation (surfG1D.py:223-399)
# --------------------------------------------------------------------------- #
def chain1d_g(E, alpha, Salpha, beta, Sbeta, eta, conv=1e-5, relFactor=0.1,
              max_iter=2000, g_init=None, force_iters=None):
    """surfG1D.py:223-295.

    A=(E+i eta)Sa-a, B=(E+i eta)Sb-b, B^H=conj(B).T (conjugates a complex E
    too, :262).  State (count, diff, g) starts at (0, inf, inv(A)) (:287).
    Body (:271-284): g_new=inv(A-B g B^H); diff=max(|g_new-g|/max(|g_new|,1e-12));
    g <- r g_new + (1-r) g; loop while diff>conv and count<2000.  Returns the
    MIXED g.  ``g_init`` overrides the start (the reference's own test,
    tests/test_surface_green_jit.py:47-68, starts from zeros); ``force_iters``
    runs exactly that many sweeps regardless of diff (parity at fixed trip
    count).  Returns (g, count, diff)."""
    A = (E + 1j * eta) * Salpha - alpha
    B = (E + 1j * eta) * Sbeta - beta
    Bd = B.conj().T
    g = inv(A) if g_init is None else np.array(g_init, dtype=complex)
    count = 0
    diff = np.inf
    while True:
        if force_iters is not None:
            if count >= force_iters:
                break
        elif not (diff > conv and count < max_iter):
            break
        g_new = inv(A - B @ g @ Bd)
        diff = np.max(np.abs(g_new - g) / np.maximum(np.abs(g_new), 1e-12))
        g = g_new * relFactor + g * (1 - relFactor)
        count += 1
    return g, count, diff


def chain1d_sigma_block(E, tau, Stau, g):
    """surfG1D.py:369-371 -- t = E Stau - tau (no eta); t g t^H."""
    t = E * Stau - tau
    return t @ g @ t.conj().T


class Chain1DSigma:
    """Fully specified ('pattern c', surfG1D.py:26-31) 1-D chain provider with the
    reference's sigma/sigmaTot protocol (surfG1D.py:344-399)."""
    def __init__(self, F, S, indsList, taus, staus, alphas, aOverlaps, betas, bOverlaps,
                 eta=1e-6, conv=1e-5, relFactor=0.1, force_iters=None):
        self.F = np.asarray(F); self.S = np.asarray(S)
        self.indsList = [np.asarray(i) for i in indsList]
        self.tauList = taus; self.stauList = staus
        self.aList = alphas; self.aSList = aOverlaps
        self.bList = betas; self.bSList = bOverlaps
        self.eta = eta; self.conv = conv; self.relFactor = relFactor
        self.force_iters = force_iters
        self.num_contacts = len(indsList)
        self.last_iters = {}

    def g(self, E, i):
        g, count, diff = chain1d_g(E, self.aList[i], self.aSList[i], self.bList[i],
                                   self.bSList[i], self.eta, self.conv, self.relFactor,
                                   force_iters=self.force_iters)
        self.last_iters[(complex(E), i)] = (count, diff)
        return g

    def sigma_block(self, E, i):
        return chain1d_sigma_block(E, self.tauList[i], self.stauList[i], self.g(E, i))

    def sigma(self, E, i, conv=None):
        out = np.zeros(self.F.shape, dtype=complex)
        inds = self.indsList[i]
        out[np.ix_(inds, inds)] += self.sigma_block(E, i)
        return out

    def sigmaTot(self, E, conv=None):
        out = np.zeros(self.F.shape, dtype=complex)
        for i in range(self.num_contacts):
            out = out + self.sigma(E, i)
        return out


def chain1d_sigma(E, N, inds, tau, Stau, alpha, Salpha, beta, Sbeta, eta, **kw):
    """surfG1D.py:344-373 -- scatter-add t g t^H into an N x N zero matrix."""
    g, count, diff = chain1d_g(E, alpha, Salpha, beta, Sbeta, eta, **kw)
    out = np.zeros((N, N), dtype=complex)
    out[np.ix_(inds, inds)] += chain1d_sigma_block(E, tau, Stau, g)
    return out, count, diff


def chain1d_sigma_total(E, N, contacts, eta, **kw):
    """surfG1D.py:375-399 -- sum over contacts; ``contacts`` is a list of dicts
    with keys inds,tau,Stau,alpha,Salpha,beta,Sbeta."""
    out = np.zeros((N, N), dtype=complex)
    for c in contacts:
        s, _, _ = chain1d_sigma(E, N, c["inds"], c["tau"], c["Stau"], c["alpha"],
                                c["Salpha"], c["beta"], c["Sbeta"], eta, **kw)
        out = out + s
    return out


# --------------------------------------------------------------------------- #
# Bethe lattice (surfGBethe.py:958-1136, 479-575)
# --------------------------------------------------------------------------- #
def bethe_sigmaK(E, H, Slist, Vlist, eta, conv=1e-5, mix=0.5, max_iter=1000,
                 force_iters=None):
    """surfGBethe.py:958-1030 -- bulk self-energies for the 12 FCC directions.

    sigma_k = -i I (k<12); A = (E - i eta) I - H (note the MINUS, :995).  One
    sweep: Sigma_tot = sum_k sigma_k is frozen for the sweep, but sigma is
    updated Gauss-Seidel style, so sigma[(k+6)%12] already holds this sweep's
    value when pair index < k (:1009-1014).  diff = max|s - s_old| / max|s_old|.
    Returns (sigmaK[12,9,9], count, diff)."""
    NN = len(Slist)
    sig = np.array([np.eye(DIM) * -1j for _ in range(NN)], dtype=complex)
    z = E - eta * 1j
    A = z * np.eye(DIM) - H
    count, diff = 0, np.inf
    while True:
        if force_iters is not None:
            if count >= force_iters:
                break
        elif not (diff > conv and count < max_iter):
            break
        old = sig.copy()
        tot = np.sum(sig, axis=0)
        for k in range(NN):
            pk = (k + 6) % 12
            gk = inv(A - tot + sig[pk])
            B = z * Slist[k] - Vlist[k]
            sig[k] = mix * (B @ gk @ B.conj().T) + (1 - mix) * old[k]
        diff = np.max(np.abs(sig - old)) / np.max(np.abs(old))
        count += 1
    return sig, count, diff


def bethe_sigma_surface(E, H, Slist, Vlist, eta, conv=1e-5, mix=0.5, max_iter=1000,
                        force_iters=None, sigK=None):
    """surfGBethe.py:1032-1108 -- surface self-energies (first 9 directions).

    Starts from sigmaK[:9]; per sweep ONE inverse g = inv(A - sum_{k<9} s_k), then
    the six in-plane directions {0,1,2,6,7,8} get s_k <- mix B_k g B_k^H +
    (1-mix) s_k_old.  Returns (sigSurf[9,9,9], count, diff, countK)."""
    countK = None
    if sigK is None:
        sigK, countK, _ = bethe_sigmaK(E, H, Slist, Vlist, eta, conv, mix, max_iter,
                                       force_iters=force_iters)
    s = np.array(sigK[:9], dtype=complex)
    z = E - eta * 1j
    A = z * np.eye(DIM) - H
    plane = [0, 1, 2, 6, 7, 8]
    count, diff = 0, np.inf
    while True:
        if force_iters is not None:
            if count >= force_iters:
                break
        elif not (diff > conv and count < max_iter):
            break
        old = s.copy()
        g = inv(A - np.sum(s, axis=0))
        for k in plane:
            B = z * Slist[k] - Vlist[k]
            s[k] = mix * (B @ g @ B.conj().T) + (1 - mix) * old[k]
        diff = np.max(np.abs(s - old)) / np.max(np.abs(old))
        count += 1
    return s, count, diff, countK


def bethe_atom_sigma(sigSurf, nInds):
    """surfGBethe.py:523-527 -- sum of the 9 surface directions minus those
    attached to device neighbours.  PINNED for neighbour lists inside 0..8 (the
    reference's numpy twin surfG3D.py:417-433, tests/golden/ref_bethe.npz asm_*).
    Outside that range the jax indexing rules of ``sigSurf[neighbor_idx]`` on the
    length-9 array are followed (jax docs, "out-of-bounds indexing": retrieval
    clamps; a negative index first wraps like numpy's): -9..-1 -> 0..8, anything
    still outside clamps to 0 / 8.  Unverified against executed reference code
    (jax is not installed); SURVEY.md section 8 a16."""
    out = np.sum(sigSurf[:9], axis=0)
    for nb in nInds:
        nb = int(nb)
        if nb < 0:
            nb += 9
        out = out - sigSurf[min(max(nb, 0), 8)]
    return out


def bethe_contact_sigma(E, N, atom_inds, atom_nInds, H, Slist, Vlist, eta,
                        Xi=None, spin='r', sigSurf=None, **kw):
    """surfGBethe.py:479-542 -- per-contact N x N (or 2N x 2N) self-energy:
    sig[ix_(Finds,Finds)] = sigma_atom (SET, not add); optional Xi sig Xi when
    the .bethe overlap parameter sss == 0 (:530-533); spin kron (:536-539).
    ``sigSurf``: given [9,9,9] surface self-energies (pinning the assembly against
    the reference's numpy twin, which is handed the same set; the Xi branch of the
    twin calls an undefined helper and cannot be pinned)."""
    if sigSurf is None:
        sigSurf, count, diff, countK = bethe_sigma_surface(E, H, Slist, Vlist, eta, **kw)
    sig = np.zeros((N, N), dtype=complex)
    for nInds, Finds in zip(atom_nInds, atom_inds):
        Finds = np.asarray(Finds)
        sig[np.ix_(Finds, Finds)] = bethe_atom_sigma(sigSurf, nInds)
    if Xi is not None:
        sig = Xi @ sig @ Xi
    if spin in ('u', 'ro'):
        sig = np.kron(np.eye(2), sig)
    elif spin == 'g':
        sig = np.kron(sig, np.eye(2))
    return sig


def bethe_cluster_sigma_total(E, H, Slist, Vlist, eta, sigK=None, **kw):
    """surfGBethe.py:1129-1136 -- 117 x 117 block-diagonal self-energy of the
    13-site cluster used for the contact Fermi level: block k (k<12) holds
    Sigma_tot - sigma_{(k+6)%12}; the centre block (last) stays zero.
    ``sigK``: given bulk self-energies (pinning against the reference's numpy twin,
    surfG3D.py:998-1031, which is handed the same set)."""
    if sigK is None:
        sigK, _, _ = bethe_sigmaK(E, H, Slist, Vlist, eta, **kw)
    NN = len(Slist)
    tot = np.sum(sigK, axis=0)
    out = np.zeros(((NN + 1) * DIM, (NN + 1) * DIM), dtype=complex)
    for k in range(NN):
        pk = (k + 6) % 12
        out[k * DIM:(k + 1) * DIM, k * DIM:(k + 1) * DIM] = tot - sigK[pk]
    return out


# --------------------------------------------------------------------------- #
# analytic density for constant self-energies (density.py:276-329): an independent check of the
# contour orientation and prefactors of the grid code
# --------------------------------------------------------------------------- #
def density_analytic(V, Vc, D, Gam, Emin, mu):
    """density.py:276-329, operation for operation (np.emath.log, broadcast by stacking rows)."""
    Nd = len(V)
    DD = np.array([D for i in range(Nd)]).T
    logmat = np.array([np.emath.log(1 - (mu / D)) for i in range(Nd)]).T
    logmat2 = np.array([np.emath.log(1 - (Emin / D)) for i in range(Nd)]).T
    invmat = 1 / (2 * np.pi * (DD - DD.conj().T))
    pref2 = logmat - logmat.conj().T
    pref3 = logmat2 - logmat2.conj().T
    prefactor = np.multiply(invmat, (pref2 - pref3))
    Gammam = Vc.conj().T @ Gam @ Vc
    prefactor = np.multiply(prefactor, Gammam)
    return V @ prefactor @ V.conj().T


def density_analytic_from_system(F, S, sig1, sig2, Emin, mu):
    """scf.py:553-575: P = X density(V, Vc, D, X Gamma X, Emin, mu) X for constant contacts (eV)."""
    from scipy.linalg import fractional_matrix_power
    X = np.array(fractional_matrix_power(S, -0.5))
    Fbar = X @ (F + sig1 + sig2) @ X
    Gam = X @ (1j * (sig1 - sig1.conj().T) + 1j * (sig2 - sig2.conj().T)) @ X
    D, V = np.linalg.eig(Fbar)
    Vc = np.linalg.inv(V.conj().T)
    return X @ density_analytic(V, Vc, D, Gam, Emin, mu) @ X


# --------------------------------------------------------------------------- #
# grid / weight bookkeeping (density.py, transport.py)
# --------------------------------------------------------------------------- #
def fermi(E, mu, T):
    """density.py:64-86.  T==0 -> (E<=mu)*1, which on complex E is numpy's
    lexicographic complex ordering."""
    kT = kB * T
    if kT == 0:
        return (E <= mu) * 1
    return 1 / (np.exp((E - mu) / kT) + 1)


def ant_points(N):
    """density.py:88-119 -- ANT-modified Gauss-Chebyshev nodes/weights."""
    k = np.arange(1, N + 1, 2)
    theta = k * np.pi / (2 * N)
    xs = np.sin(theta)
    xcc = np.cos(theta)
    x = 1.0 + 0.21220659078919378103 * xs * xcc * (3 + 2 * xs * xs) - k / (N)
    x = np.concatenate((x, -1 * x))
    w = xs ** 4 * 16.0 / (3 * (N))
    w = np.concatenate((w, w))
    return x, w


def real_axis_grid(Emin, mu, N, T):
    """density.py:418-427 (densityRealN) -> (Elist, weights); result prefactor is
    -Im(.)/pi (:436)."""
    from scipy.special import roots_legendre
    kT = kB * T
    Emax = mu + N_KT * kT
    mid = (Emax - Emin) / 2
    x, w = roots_legendre(N)
    x = np.real(x)
    Elist = mid * (x + 1) + Emin
    weights = mid * w * fermi(Elist, mu, T)
    return Elist, weights


def bias_window_grid(mu1, mu2, N, T):
    """density.py:519-534 (densityGridN) -> (energies, weights); prefactor 1/(2 pi)."""
    from scipy.special import roots_legendre
    kT = kB * T
    muLo = min(mu1, mu2)
    muHi = max(mu1, mu2)
    dInt = np.sign(mu2 - mu1)
    Emax = muHi + N_KT * kT
    Emin = muLo - N_KT * kT
    mid = (Emax - Emin) / 2
    x, w = roots_legendre(N)
    x = np.real(x)
    energies = mid * (x + 1) + Emin
    dfermi = fermi(energies, muHi, T) - fermi(energies, muLo, T)
    weights = mid * w * dfermi * dInt
    return energies, weights


def contour_grid(Emin, mu, N, T, method='ant'):
    """density.py:697-722 (densityComplexN) -> (Elist, weights) on the upper
    semicircle z = c + r e^{i theta}; prefactor +Im(.)/pi (:748)."""
    from scipy.special import roots_legendre
    broadening = 10 * kB * T
    Emax = mu - broadening
    center = (Emin + Emax) / 2
    r = (Emax - Emin) / 2
    if method == 'legendre':
        x, w = roots_legendre(N)
    elif method == 'chebyshev':
        k = np.arange(1, N + 1)
        x = np.cos(k * np.pi / (N + 1))
        w = (np.pi / (N + 1)) * (np.sin(k * np.pi / (N + 1)) ** 2) / np.sqrt(1 - (x ** 2))
    elif method == 'ant':
        x, w = ant_points(N)
    else:
        x = np.linspace(-1, 1, N)
        w = 2 * np.ones(N) / N
    theta = np.pi / 2 * (x + 1)
    Elist = center + r * np.exp(1j * theta)
    dz = 1j * r * np.exp(1j * theta)
    weights = (np.pi / 2) * w * fermi(Elist, mu, T) * dz
    return Elist, weights


def broadening_grid(mu, N, T, method='ant'):
    """density.py:730-742 -- the extra real-axis segment [mu-10kT, mu+10kT] with
    Nbroad = N//8 points used when T>0."""
    from scipy.special import roots_legendre
    broadening = 10 * kB * T
    Nbroad = int(N // 8)
    if method in ('legendre', 'chebyshev', 'ant'):
        x, w = roots_legendre(Nbroad)
    else:
        x = np.linspace(-1, 1, Nbroad)
        w = 2 * np.ones(Nbroad) / Nbroad
    Elist = broadening * x + mu
    weights = broadening * w * fermi(Elist, mu, T)
    return Elist, weights


def adaptive_ant(computePoint, tol=1e-4, maxN=1000, record=None):
    """density.py:211-273 -- nested ANT levels N=2,6,18,...; each level evaluates
    only the NEW nodes and rescales the previous value by the nested-weight
    ratio.  ``record`` (a list) receives (N, new-node x, new-node w, ratio)."""
    prev_x = None
    prev_sumW = None
    P = None
    new_P = None
    N = 2
    maxDP = 1e10
    while N <= maxN:
        x, w = ant_points(N)
        if prev_x is None:
            P = computePoint(x[0:2], w[0:2])
            if record is not None:
                record.append((N, x[0:2].copy(), w[0:2].copy(), None))
        else:
            old_mask = np.isin(np.round(x, 14), np.round(prev_x, 14))
            assert int(old_mask.sum()) == prev_x.size, "Old nodes mismatch"
            ratio = float(np.sum(w[old_mask]) / prev_sumW)
            new_mask = ~old_mask
            new_P = P * ratio
            new_P += computePoint(x[new_mask], w[new_mask])
            if record is not None:
                record.append((N, x[new_mask].copy(), w[new_mask].copy(), ratio))
            maxDP = np.max(np.abs(new_P - P))
            P = new_P.copy()
            if maxDP < tol:
                return new_P
        prev_x = x
        prev_sumW = float(np.sum(w))
        N *= 3
    return new_P


def current_grid(fermi_E, qV, T=0.0, dE=0.001):
    """transport.py:652-675 -- np.arange window (end-exclusive), dE sign follows qV."""
    if qV < 0:
        dE = -1 * abs(dE)
    else:
        dE = abs(dE)
    muL = fermi_E - qV / 2
    muR = fermi_E + qV / 2
    if T == 0:
        grid = np.arange(muL, muR, dE)
    else:
        spread = np.sign(dE) * N_KT * kB * T
        grid = np.arange(muL - spread, muR + spread, dE)
    return grid, muL, muR


def current_from_transmission(transmissions, energies, muL, muR, T=0.0, spin='r'):
    """transport.py:692-720 -- trapezoid(T |df|) * e/h, x2 for restricted spin."""
    from scipy.integrate import trapezoid
    transmissions = np.asarray(transmissions)
    if T == 0:
        cur = eoverh * trapezoid(transmissions, energies)
    else:
        dfermi = np.abs(1 / (np.exp((energies - muR) / (kB * T)) + 1) -
                        1 / (np.exp((energies - muL) / (kB * T)) + 1))
        cur = eoverh * trapezoid(transmissions * dfermi, energies)
    if spin == 'r':
        cur *= 2
    return cur
