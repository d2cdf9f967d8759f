"""
Energy-independent self-energy provider (drop-in for gauNEGF/surfGTester.py:15-152).

``sigma`` / ``sigmaTot`` return the stored host matrices (they are inputs, not
computed quantities); ``_negf_lower`` hands them to the engine once as a CONST
provider, after which every energy point of an integral is assembled and inverted
on the GPU without touching the host again.
"""
import numpy as np

from .config import SURFACE_GREEN_CONVERGENCE
from .matTools import formSigma


class surfGTest:
    def __init__(self, Fock, Overlap, indsList, sig1=None, sig2=None):
        self.F = Fock
        self.S = Overlap
        self.N = len(Fock)
        self.indsList = indsList
        if sig1 is not None:
            # surfGTester.py:84-89: both contacts use sig1 unless sig2 is given
            self.sig = [formSigma(indsList[0], sig1, self.N, self.S),
                        formSigma(indsList[1], sig1 if sig2 is None else sig2, self.N, self.S)]
        else:
            # documented default "-0.05j on contact orbitals" (surfGTester.py:52).  The
            # reference's own default branch (:91-92) aliases one array for both
            # contacts and assigns an N x N diagonal into the contact block, which only
            # works when a contact spans all orbitals; the documented intent is built.
            self.sig = []
            for inds in indsList[:2]:
                s = np.zeros((self.N, self.N), dtype=complex)
                idx = np.asarray(list(inds), dtype=int)
                s[idx, idx] = -0.05j
                self.sig.append(s)
        self._lowered = {}

    def sigma(self, E, i, conv=SURFACE_GREEN_CONVERGENCE):
        return self.sig[i]

    def sigmaTot(self, E, conv=SURFACE_GREEN_CONVERGENCE):
        total = np.zeros((self.N, self.N), dtype=complex)
        for i in range(len(self.indsList)):
            total += self.sigma(E, i, conv)
        return total

    def setF(self, F, mu1=None, mu2=None):
        self.F = F

    # ---- engine lowering ---------------------------------------------------
    @property
    def num_contacts(self):
        return len(self.indsList)

    def _negf_lower(self, engine):
        key = (id(engine), getattr(engine, "generation", 0))
        if key not in self._lowered:
            self._release()
            mats = [self.sig[i] for i in range(len(self.indsList))]
            self._lowered[key] = (engine, engine.sigma_const(mats))
        return self._lowered[key][1]

    def _negf_spin_split(self, N):
        """Two N x N constant providers for the diagonal spin blocks of a 2N x 2N system, or None when a
        self-energy matrix couples the blocks (integrate.py's block-diagonal fast path)."""
        if self.N != 2 * N:
            return None
        # the halves (and the CONST providers they lower to) are kept with the object: GrInt / GrLessInt ask for
        # them on every call, and creating / freeing device providers each time dominated small integrals
        from .engine import fingerprint
        key = (N, tuple((id(sg), sg.shape, fingerprint(sg)) for sg in self.sig))
        cached = getattr(self, "_split_cache", None)
        if cached is not None and cached[0] == key:
            for h in cached[1] or ():
                h.F = np.asarray(self.F)[h._sl, h._sl]; h.S = np.asarray(self.S)[h._sl, h._sl]
            return cached[1]
        for sg in self.sig:
            if np.any(sg[:N, N:]) or np.any(sg[N:, :N]):
                self._split_cache = (key, None)
                return None
        halves = []
        for sl in (slice(0, N), slice(N, 2 * N)):
            h = object.__new__(surfGTest)
            h.F = np.asarray(self.F)[sl, sl]; h.S = np.asarray(self.S)[sl, sl]; h.N = N
            h.indsList = self.indsList
            h.sig = [np.ascontiguousarray(sg[sl, sl]) for sg in self.sig]
            h._lowered = {}
            h._sl = sl
            halves.append(h)
        self._split_cache = (key, halves)
        return halves

    def _release(self):
        # handles are never reused by the library: freeing one that a change of the matrix dimension
        # already dropped is a no-op there
        for eng, h in self._lowered.values():
            eng.sigma_free(h)
        self._lowered.clear()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
