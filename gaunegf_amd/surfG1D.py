"""
1-D chain contact self-energies -- drop-in for ``gauNEGF.surfG1D.surfG``
(gauNEGF/surfG1D.py:12-399).

The class keeps the reference's constructor patterns and attribute names
(``aList, aSList, bList, bSList, tauList, stauList, indsList, F, S, X, eta``) because
host code reads them (density.py:1037-1040).  The decimation fixed point itself
(surfG1D.py:223-295) and ``Sigma = t g t^H`` (:344-373) run on the GPU: the object
lowers itself to a CHAIN1D provider of libnegf_hip.so and ``g / sigma / sigmaTot``
are thin calls into it.  Inside GrInt/GrLessInt the provider is used directly on the
device, one workgroup per (energy, contact), and Sigma never visits the host.
"""
import numpy as np

from .config import ETA, SURFACE_GREEN_CONVERGENCE, SURFACE_RELAXATION_FACTOR, SURFACE_GREEN_MAX_ITER


def fractional_matrix_power(S, power):
    """S^p of a Hermitian matrix through its eigen-decomposition, eigenvalues
    clamped at 1e-16 (gauNEGF/utils.py:12-48).  Setup-time host work."""
    vals, vecs = np.linalg.eigh(np.asarray(S))
    vals = np.maximum(vals, 1e-16)
    return vecs @ np.diag(np.power(vals, power)) @ vecs.conj().T


class surfG:
    def __init__(self, Fock, Overlap, indsList, taus=None, staus=None, alphas=None, aOverlaps=None,
                 betas=None, bOverlaps=None, eta=ETA):
        self.F = np.array(Fock)
        self.S = np.array(Overlap)
        self.X = np.array(fractional_matrix_power(Overlap, -0.5))
        self.indsList = [np.array(inds) for inds in indsList]
        self.num_contacts = len(indsList)
        self._version = 0
        self._lowered = {}
        self.force_iters = -1          # >= 0: run exactly that many sweeps (parity tests)

        # coupling: index lists (taken from F/S) or explicit matrices   (surfG1D.py:131-143)
        if taus is None:
            taus = [self.indsList[-1], self.indsList[0]]
        taus = [np.array(t) for t in taus]
        if np.ndim(taus[0]) == 1:
            self.tauFromFock = True
            self.tauInds = taus
            self._coupling_from_fock()
        else:
            self.tauFromFock = False
            self.tauList = [np.array(t) for t in taus]
            self.stauList = [np.array(s) for s in staus]

        # lead unit cell: from F/S or fully specified                   (surfG1D.py:145-152)
        if alphas is None:
            self.contactFromFock = True
            self.setContacts()
        else:
            self.contactFromFock = False
            self.setContacts(alphas, aOverlaps, betas, bOverlaps)
            self.fermiList = [None] * len(indsList)
        self.eta = eta

    def _coupling_from_fock(self):
        t, ends = self.tauInds, (self.indsList[0], self.indsList[-1])
        self.tauList = [self.F[np.ix_(t[k], ends[k])] for k in range(2)]
        self.stauList = [self.S[np.ix_(t[k], ends[k])] for k in range(2)]

    def setContacts(self, alphas=None, aOverlaps=None, betas=None, bOverlaps=None):
        """surfG1D.py:167-221."""
        if self.contactFromFock:
            self.aList = [np.array(self.F[np.ix_(i, i)]) for i in self.indsList]
            self.aSList = [np.array(self.S[np.ix_(i, i)]) for i in self.indsList]
            self.bList = [np.array(t) for t in self.tauList]
            self.bSList = [np.array(s) for s in self.stauList]
        else:
            self.aList = [np.array(a) for a in alphas]
            self.aSList = [np.array(a) for a in aOverlaps]
            self.bList = [np.array(b) for b in betas]
            self.bSList = [np.array(b) for b in bOverlaps]
        self._version += 1

    def setF(self, F, mu1=None, mu2=None):
        """surfG1D.py:297-342.  As in the reference, a new F refreshes the coupling
        blocks (tau) but not the lead cell (alpha/beta) extracted at construction."""
        self.F = np.array(F)
        if self.tauFromFock:
            t, inds = self.tauInds, self.indsList
            self.F[np.ix_(inds[0], inds[0])] = self.F[np.ix_(t[0], t[0])].copy()
            self.F[np.ix_(inds[-1], inds[-1])] = self.F[np.ix_(t[1], t[1])].copy()
            self._coupling_from_fock()
        if not self.contactFromFock:
            if self.fermiList[0] is None:
                self.fermiList[0] = mu1
                self.fermiList[-1] = mu2
            else:
                # intended behaviour of surfG1D.py:331-342 (the reference calls .at on a
                # Python list there and would raise): shift the lead by the change of mu
                for i, mu in zip([0, -1], [mu1, mu2]):
                    old = self.fermiList[i]
                    if old is not None and mu is not None and old != mu:
                        d = mu - old
                        self.aList[i] = self.aList[i] + d * np.eye(len(self.aList[i]))
                        self.bList[i] = self.bList[i] + d * self.bSList[i]
                        self.fermiList[i] = mu
        self._version += 1

    # ---- engine lowering ---------------------------------------------------
    def _engine(self):
        from .engine import get_engine
        eng = get_engine()
        if eng.n != self.F.shape[0]:
            eng.set_system(self.F, self.S)
        return eng

    def _contact_ids(self):
        return list(range(self.num_contacts))

    def _negf_lower(self, engine, conv=SURFACE_GREEN_CONVERGENCE, relFactor=SURFACE_RELAXATION_FACTOR,
                    identity_tau=False):
        """CHAIN1D provider handle for (conv, relFactor); cached until setF/setContacts.
        ``identity_tau`` builds the variant with t = I (tau = -I, Stau = 0), whose
        'self-energy' is the surface Green's function g itself."""
        key = (id(engine), getattr(engine, "generation", 0), self._version, float(conv), float(relFactor),
               bool(identity_tau), float(self.eta), int(self.force_iters))
        if key in self._lowered:
            return self._lowered[key][1]
        if len(self._lowered) > 8:
            self._release()
        ids = self._contact_ids()
        nc = [len(self.indsList[i]) for i in ids]
        taus, staus = [], []
        for k, i in enumerate(ids):
            if identity_tau:
                taus.append(-np.eye(nc[k])); staus.append(np.zeros((nc[k], nc[k])))
            else:
                t = np.asarray(self.tauList[i]); s = np.asarray(self.stauList[i])
                if t.shape != (nc[k], nc[k]):
                    raise ValueError(f"contact {i}: coupling block {t.shape} must be {nc[k]}x{nc[k]} "
                                     "(t g t^H is added at ix_(inds, inds), surfG1D.py:372)")
                taus.append(t); staus.append(s)
        h = engine.sigma_chain1d([self.indsList[i] for i in ids],
                                 [self.aList[i] for i in ids], [self.aSList[i] for i in ids],
                                 [self.bList[i] for i in ids], [self.bSList[i] for i in ids],
                                 taus, staus, self.eta, conv, relFactor,
                                 max_iter=SURFACE_GREEN_MAX_ITER, force_iters=self.force_iters)
        self._lowered[key] = (engine, h)
        return h

    def _release(self):
        # handles are never reused by the library: freeing a stale one is a no-op there
        for eng, h in self._lowered.values():
            eng.sigma_free(h)
        self._lowered.clear()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---- reference protocol ------------------------------------------------
    def g(self, E, i, conv=SURFACE_GREEN_CONVERGENCE, relFactor=SURFACE_RELAXATION_FACTOR):
        """Surface Green's function of contact i at E (surfG1D.py:223-295)."""
        eng = self._engine()
        h = self._negf_lower(eng, conv, relFactor, identity_tau=True)
        full = eng.sigma_eval(h, i, [E], self.num_contacts)[0]
        inds = self.indsList[i]
        return full[np.ix_(inds, inds)]

    def sigma(self, E, i, conv=SURFACE_GREEN_CONVERGENCE):
        """N x N self-energy of contact i (surfG1D.py:344-373)."""
        eng = self._engine()
        h = self._negf_lower(eng, conv)
        return eng.sigma_eval(h, i, [E], self.num_contacts)[0]

    def sigmaTot(self, E, conv=SURFACE_GREEN_CONVERGENCE):
        """Sum over contacts (surfG1D.py:375-399)."""
        eng = self._engine()
        h = self._negf_lower(eng, conv)
        return eng.sigma_eval(h, None, [E], self.num_contacts)[0]

    def sigma_batch(self, Elist, i=None, conv=SURFACE_GREEN_CONVERGENCE):
        """All energies in one launch: [M,N,N] plus (iterations, converged) [M,contacts]."""
        eng = self._engine()
        h = self._negf_lower(eng, conv)
        out = eng.sigma_eval(h, i, Elist, self.num_contacts)
        return out, eng.last_iters.copy(), eng.last_converged.copy()
