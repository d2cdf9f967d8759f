"""
Bethe-lattice contact self-energies -- drop-in for ``gauNEGF.surfGBethe``
(``surfGB`` gauNEGF/surfGBethe.py:47-575, ``surfGBAt`` :830-1188).

Host side (done once, numpy): ``.bethe`` parameter file, Slater-Koster 9x9 hopping /
overlap blocks for the 12 FCC nearest-neighbour directions, contact geometry
(surface normal, attached directions per contact atom).  Device side (every energy
of every integral): the coupled bulk fixed point over 12 directions, the surface
fixed point over the 6 in-plane directions and the per-atom assembly
(surfGBethe.py:958-1108, 512-527) run in one HIP kernel, one workgroup per
(energy, contact); inside GrInt/GrLessInt the 9x9 blocks are consumed on the device.
"""
import os

import numpy as np

from .config import ETA, TEMPERATURE, SURFACE_GREEN_CONVERGENCE, FERMI_CALCULATION_TOL, ENERGY_MIN, \
    BETHE_MAX_ITER, BETHE_MIX
from .surfG1D import fractional_matrix_power

kB = 8.617e-5           # eV/Kelvin
dim = 9                 # 1 s + 3 p + 5 d orbitals per contact atom
har_to_eV = 27.211386   # eV/Hartree
Eminf = ENERGY_MIN
bohr_to_ang = 0.529177

_EXPECTED_KEYS = ['ne', 'es', 'ep', 'edd', 'edt', 'sss', 'sps', 'pps', 'ppp', 'sds', 'pds', 'pdp',
                  'dds', 'ddp', 'ddd', 'Ssss', 'Ssps', 'Spps', 'Sppp', 'Ssds', 'Spds', 'Spdp', 'Sdds',
                  'Sddp', 'Sddd']


# --------------------------------------------------------------------------- #
# parameter file and Slater-Koster blocks (host, setup time)
# --------------------------------------------------------------------------- #
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def lattice_file(name):
    """Path (without the ``.bethe`` extension, as ``latFile`` wants it) of a parameter table shipped with the package:
    ``Au``, ``Au2`` -- the reference keeps them in its checkout root and opens them relative to the working directory."""
    return os.path.join(DATA_DIR, name)


def read_bethe_params(filename):
    """Parse ``<filename>.bethe`` (``key = value`` lines, 25 keys) into
    (ne, Edict[eV], Vdict[eV], Sdict, H0) -- surfGBethe.py:301-355.  A bare name that is not found relative to the
    working directory (where the reference looks) is looked up among the tables shipped in gaunegf_amd/data."""
    params = {}
    if not os.path.exists(filename + '.bethe') and not os.path.dirname(filename) and \
            os.path.exists(lattice_file(filename) + '.bethe'):
        filename = lattice_file(filename)
    with open(filename + '.bethe', 'r') as f:
        for line in f:
            if not line.strip():
                continue
            key, value = line.replace(' ', '').split('=')
            params[key] = float(value)
    assert len(params) == len(_EXPECTED_KEYS) and set(params) == set(_EXPECTED_KEYS), \
        f"Error reading file: Found Bethe parameters: {list(params.keys())}, expected: {_EXPECTED_KEYS}"
    ne = params['ne']
    Edict = {k[1:]: params[k] * har_to_eV for k in params if k.startswith('e')}
    Sdict = {k[1:]: params[k] for k in params if k.startswith('S')}
    Vdict = {k: params[k] * har_to_eV for k in params if not k.startswith('e') and not k.startswith('S')}
    hdiag = [Edict['s']] + [Edict['p']] * 3 + [Edict['dd']] + [Edict['dt']] * 2 + [Edict['dd'], Edict['dt']]
    return ne, Edict, Vdict, Sdict, np.diag(np.array(hdiag))


def _bond_frame_block(P):
    """9x9 two-centre block for a bond along +z in the order
    [s, px, py, pz, d3z2-r2, dxz, dyz, dx2-y2, dxy] (surfGBethe.py:386-418)."""
    M = np.zeros((dim, dim))
    M[0, 0] = P['sss']
    M[0, 3], M[3, 0] = P['sps'], -P['sps']
    M[1, 1] = M[2, 2] = P['ppp']
    M[3, 3] = P['pps']
    M[0, 4] = M[4, 0] = P['sds']
    M[1, 5], M[5, 1] = P['pdp'], -P['pdp']
    M[2, 6], M[6, 2] = P['pdp'], -P['pdp']
    M[3, 4], M[4, 3] = P['pds'], -P['pds']
    M[4, 4] = P['dds']
    M[5, 5] = M[6, 6] = P['ddp']
    M[7, 7] = M[8, 8] = P['ddd']
    return M


def _rotation(dirCosines):
    """Orbital rotation taking the +z bond frame to direction (x,y,z)
    (surfGBethe.py:420-473; d block as in ANT.Gaussian)."""
    x, y, z = dirCosines
    th = np.arccos(z)
    ph = np.arctan2(y, x)
    ct, st, cp, sp = np.cos(th), np.sin(th), np.cos(ph), np.sin(ph)
    s2t, c2t, c2p, s2p = np.sin(2 * th), np.cos(2 * th), np.cos(2 * ph), np.sin(2 * ph)
    r3 = np.sqrt(3)
    tr = np.zeros((dim, dim))
    tr[0, 0] = 1.0
    tr[1:4, 1:4] = [[ct * cp, -sp, st * cp],
                    [ct * sp, cp, st * sp],
                    [-st, 0, ct]]
    d10 = r3 * s2t * cp / 2
    d20 = r3 * s2t * sp / 2
    tr[4:9, 4:9] = [
        [(3 * z ** 2 - 1) / 2, -r3 * s2t / 2, 0.0, r3 * st ** 2 / 2, 0.0],
        [d10, c2t * cp, -ct * sp, -d10 / r3, st * sp],
        [d20, c2t * sp, ct * cp, -d20 / r3, -st * cp],
        [r3 * st ** 2 * c2p / 2, s2t * c2p / 2, -st * s2p, (1 + ct ** 2) * c2p / 2, -ct * s2p],
        [r3 * st ** 2 * s2p / 2, s2t * s2p / 2, st * c2p, (1 + ct ** 2) * s2p / 2, ct * c2p],
    ]
    return tr


def construct_sk_matrix(Mdict, dirCosines):
    """Slater-Koster block for a bond along ``dirCosines`` (surfGBethe.py:357-477)."""
    tr = _rotation(np.asarray(dirCosines, dtype=float))
    return tr @ _bond_frame_block(Mdict) @ tr.T


def _rodrigues(axis, angle):
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def gen_neighbors(plane_normal, first_neighbor):
    """12 FCC [111] nearest-neighbour unit vectors: 3 in-plane (60 degree steps), 3 out-of
    -plane (120 degree steps, tilted by acos(1/sqrt 3)), then their opposites at k+6
    (surfGBethe.py:220-298)."""
    plane_normal = np.asarray(plane_normal, dtype=float)
    proj = first_neighbor - np.dot(first_neighbor, plane_normal) * plane_normal
    first = proj / np.linalg.norm(proj)
    vecs = []
    for i in range(3):
        v = _rodrigues(plane_normal, i * np.pi / 3) @ first
        vecs.append(v / np.linalg.norm(v))
    tilt = np.arccos(1 / np.sqrt(3))
    base = np.cos(tilt) * (_rodrigues(plane_normal, np.pi / 6) @ first) + np.sin(tilt) * plane_normal
    for i in range(3):
        vecs.append(_rodrigues(plane_normal, i * 2 * np.pi / 3) @ base)
    return vecs + [-v for v in vecs[:6]]


# --------------------------------------------------------------------------- #
# single-atom Bethe lattice (surfGBethe.py:830-1188)
# --------------------------------------------------------------------------- #
class surfGBAt:
    """One Bethe-lattice atom: onsite H (9x9), 12 overlap and hopping blocks.  ``F``/``S``
    are the 117 x 117 matrices of the 13-site cluster (12 neighbours, then the centre)
    used by the contact Fermi-level search."""

    def __init__(self, H, Slist, Vlist, eta, T=TEMPERATURE):
        H = np.asarray(H)
        assert H.shape == (dim, dim), f"Error with H dim, should be {dim}x{dim}"
        for S, V in zip(Slist, Vlist):
            assert np.shape(S) == (dim, dim), f"Error with S dim, should be {dim}x{dim}"
            assert np.shape(V) == (dim, dim), f"Error with F dim, should be {dim}x{dim}"
        self.H = H
        self.Slist = [np.asarray(s) for s in Slist]
        self.Vlist = [np.asarray(v) for v in Vlist]
        self.NN = len(Slist)
        assert self.NN == 12, "Error: surfGBAt only implemented for FCC using 12 NN"
        self.eta = eta
        self.T = T
        self.fermi = None
        self.force_iters = -1
        self._version = 0
        self._lowered = {}
        self.updateH()

    def updateH(self, fermi=None):
        """Shift the lattice to a new Fermi level and rebuild the cluster matrices
        (surfGBethe.py:912-955)."""
        if fermi is not None and self.fermi is not None and fermi != self.fermi:
            d = fermi - self.fermi
            self.H = self.H + d * np.eye(dim)
            self.Vlist = [V + d * S for V, S in zip(self.Vlist, self.Slist)]
            self.fermi = fermi
        n = dim * (self.NN + 1)
        H0x = np.kron(np.eye(self.NN + 1), self.H)
        S0x = np.eye(n)
        for i in range(self.NN):
            blk = slice(i * dim, (i + 1) * dim)
            S0x[-dim:, blk] = self.Slist[i]
            S0x[blk, -dim:] = self.Slist[i].T
            H0x[-dim:, blk] = self.Vlist[i]
            H0x[blk, -dim:] = self.Vlist[i].conj().T
        self.F = H0x
        self.S = S0x
        self._version += 1

    # ---- reference protocol (device evaluation) ----------------------------
    def _eval(self, E, conv, mix, which):
        from .engine import get_engine
        eng = get_engine()
        scalar = np.ndim(E) == 0
        out = eng.bethe_raw(self.H, self.Slist, self.Vlist, self.eta, conv, np.atleast_1d(E), which,
                            mix=mix, max_iter=BETHE_MAX_ITER, force_iters=self.force_iters)
        self.last_iters = eng.last_iters.copy()
        self.last_converged = eng.last_converged.copy()
        return out[0] if scalar else out

    def sigmaK(self, E, conv=SURFACE_GREEN_CONVERGENCE, mix=BETHE_MIX):
        """Bulk self-energies of the 12 lattice directions, [12,9,9] (surfGBethe.py:958-1030);
        an array of energies gives [M,12,9,9] from one launch."""
        return self._eval(E, conv, mix, 1)

    def sigma(self, E, conv=SURFACE_GREEN_CONVERGENCE, mix=BETHE_MIX):
        """Surface self-energies of the first 9 directions, [9,9,9] (surfGBethe.py:1032-1108)."""
        return self._eval(E, conv, mix, 2)

    def sigmaTot(self, E, conv=SURFACE_GREEN_CONVERGENCE):
        """117 x 117 self-energy of the 13-site cluster: block k (k<12) holds
        Sigma_tot - sigma_{(k+6)%12}, the centre block is empty (surfGBethe.py:1129-1136)."""
        sigK = self.sigmaK(E, conv)
        tot = np.sum(sigK, axis=0)
        sig = np.zeros(((self.NN + 1) * dim, (self.NN + 1) * dim), dtype=complex)
        for k in range(self.NN):
            blk = slice(k * dim, (k + 1) * dim)
            sig[blk, blk] = tot - sigK[(k + 6) % 12]
        return sig

    def sigmaTot_batch(self, Elist, conv=SURFACE_GREEN_CONVERGENCE):
        """[M,117,117] cluster self-energies from ONE launch of the Bethe kernel (used by the
        grid integrals of the contact Fermi-level search)."""
        sigK = self.sigmaK(np.asarray(Elist), conv)                     # [M,12,9,9]
        tot = np.sum(sigK, axis=1)
        out = np.zeros((sigK.shape[0], (self.NN + 1) * dim, (self.NN + 1) * dim), dtype=complex)
        for k in range(self.NN):
            blk = slice(k * dim, (k + 1) * dim)
            out[:, blk, blk] = tot - sigK[:, (k + 6) % 12]
        return out

    def DOS(self, E):
        """Bulk DOS of the lattice, -Im Tr G / pi with G = inv((E - i eta) - H - sum sigma(E))
        (surfGBethe.py:1139-1155)."""
        from .engine import get_engine
        eng = get_engine()
        eng.set_system(self.H, np.eye(dim))
        h = eng.sigma_precomputed(np.sum(self.sigma(E), axis=0)[None])
        try:
            return float(eng.dos(h, [E - 1j * self.eta], per_site=False)[0])
        finally:
            eng.sigma_free(h)

    def calcFermi(self, ne, fGuess=5, tol=FERMI_CALCULATION_TOL):
        """Contact Fermi level from the electron count (surfGBethe.py:1158-1188)."""
        from .density import getFermiContact
        self.fermi = getFermiContact(self, ne, tol, ENERGY_MIN, 1000, T=self.T, nOrbs=dim)
        return self.fermi

    def setF(self, F, mu1, mu2):
        pass        # lattice properties are intrinsic (surfGBethe.py:1110-1126)


# --------------------------------------------------------------------------- #
# device with Bethe-lattice contacts (surfGBethe.py:47-575)
# --------------------------------------------------------------------------- #
class surfGB:
    """Self-energies of FCC [111] Bethe-lattice contacts.

    ``surfGB(F, S, contacts, bar, latFile, spin, eta, T)`` follows the reference: ``bar``
    needs ``ibfatm`` (atom of each basis function, 1-based), ``ibftyp`` (orbital type
    codes) and ``c`` (coordinates in Bohr, flat).  ``surfGB.from_arrays`` builds the same
    object from plain arrays, and ``fermi=`` skips the contact Fermi-level search."""

    def __init__(self, F, S, contacts, bar, latFile='Au', spin='r', eta=ETA, T=TEMPERATURE, fermi=None):
        orbMap = np.asarray(bar.ibfatm)[np.asarray(bar.ibfatm) > 0]
        orbTyp = np.asarray(bar.ibftyp)[np.asarray(bar.ibfatm) > 0]
        c = np.asarray(bar.c, dtype=float)
        coords = c.reshape(-1, 3) * bohr_to_ang
        self._setup(F, S, contacts, orbMap, orbTyp, coords, latFile, spin, eta, T, fermi)

    @classmethod
    def from_arrays(cls, F, S, contacts, orbMap, orbTyp, coords, latFile='Au', spin='r', eta=ETA,
                    T=TEMPERATURE, fermi=None):
        """``coords`` in Angstrom [n_atoms,3]; ``orbMap`` 1-based atom id per basis function."""
        self = cls.__new__(cls)
        self._setup(F, S, contacts, np.asarray(orbMap), np.asarray(orbTyp), np.asarray(coords, dtype=float),
                    latFile, spin, eta, T, fermi)
        return self

    def _setup(self, F, S, contacts, orbMap, orbTyp, coords, latFile, spin, eta, T, fermi):
        self.cVecs, self.latVecs, self.indsLists, self.dirLists, self.nIndLists = [], [], [], [], []
        self.Xi = fractional_matrix_power(S, 0.5)
        if spin != 'r':
            self.Xi = self.Xi[::2, ::2]
        self.spin = spin
        self.N = len(orbMap)
        cList = None
        for contact in contacts:
            indsList, cList = [], []
            for atom in contact:
                inds = np.where(np.isin(orbMap, atom))[0]
                cList.append(coords[atom - 1])
                assert len(inds) == 9, f'Error: Atom {atom} has {len(inds)} basis functions, expecting 9'
                inds = inds[np.argsort(abs(orbTyp[inds]) // 1000)]
                indsList.append(inds)
            self.indsLists.append(indsList)
            cList = np.array(cList)
            # surface normal: smallest singular direction of the centred contact atoms,
            # oriented away from the molecule's centroid
            _, _, Vt = np.linalg.svd(cList - np.mean(cList, axis=0))
            contVec = Vt[-1]
            if np.dot(np.mean(cList, axis=0) - np.mean(coords, axis=0), contVec) < 0:
                contVec = -contVec
            self.cVecs.append(contVec)
            vInd = int(np.argmin(np.array([np.linalg.norm(v - cList[0]) for v in cList[1:]]))) + 1
            latVec = cList[vInd] - cList[0]
            latDist = np.linalg.norm(latVec)
            self.latVecs.append(latVec / latDist)
            nVecs1 = gen_neighbors(contVec, latVec)
            nVecs2 = gen_neighbors(contVec, -latVec)
            nIndList = []
            nVecs = list(nVecs1)
            for cpos in cList:
                nAtVecs = []
                for c2 in coords:
                    l = np.linalg.norm(c2 - cpos)
                    if 0.8 * latDist < l < 1.2 * latDist and not np.allclose(c2, cpos):
                        nAtVecs.append((c2 - cpos) / l)
                # the lattice may be rotated by 60 degrees: pick the orientation whose
                # out-of-plane directions match an actual neighbour
                nVecs = list(nVecs1)
                for vec in nAtVecs:
                    vals = np.array([np.dot(vec, d) for d in nVecs2])
                    k = int(np.argmax(vals))
                    if k in (3, 4, 5, 9, 10, 11) and vals[k] > 0.9:
                        nVecs = list(nVecs2)
                        break
                nInds = []
                for vec in nAtVecs:
                    vals = np.array([np.dot(vec, d) for d in nVecs])
                    k = int(np.argmax(vals))
                    if vals[k] > 0.9:
                        nInds.append(k)
                    else:
                        print(f'Warning: Lattice Vec #{k} mismatch, neighbor not recorded')
                nIndList.append(nInds)
            self.nIndLists.append(nIndList)
            self.dirLists.append(nVecs)

        self.readBetheParams(latFile)
        self.Slists = [[construct_sk_matrix(self.Sdict, d) for d in dl] for dl in self.dirLists]
        self.Vlists = [[construct_sk_matrix(self.Vdict, d) for d in dl] for dl in self.dirLists]
        self.gList = [surfGBAt(self.H0.copy(), Sl, Vl, eta, T) for Sl, Vl in zip(self.Slists, self.Vlists)]
        if fermi is None:
            from .density import getFermiContact
            fermi = getFermiContact(self.gList[0], self.ne / 2, FERMI_CALCULATION_TOL, ENERGY_MIN, 1000,
                                    T=T, nOrbs=dim)
        for g in self.gList:
            g.fermi = fermi
        self.cList = cList
        self.F = F
        self.S = S
        self.eta = eta
        self.force_iters = -1
        self._version = 0
        self._lowered = {}

    def readBetheParams(self, filename):
        self.ne, self.Edict, self.Vdict, self.Sdict, self.H0 = read_bethe_params(filename)

    def genNeighbors(self, plane_normal, first_neighbor):
        return gen_neighbors(plane_normal, first_neighbor)

    def constructMat(self, Mdict, dirCosines):
        return construct_sk_matrix(Mdict, dirCosines)

    @property
    def num_contacts(self):
        return len(self.indsLists)

    # ---- engine lowering ---------------------------------------------------
    def _engine(self):
        from .engine import get_engine
        eng = get_engine()
        if eng.n != self.N or self.spin != 'r':
            eng.set_system(np.zeros((self.N, self.N)), np.eye(self.N))
        return eng

    def _negf_lower(self, engine, conv=SURFACE_GREEN_CONVERGENCE):
        """BETHE provider on the N x N (spin-restricted) orbital space."""
        if engine.n != self.N:
            raise ValueError("Bethe provider is defined on the spin-restricted orbital space; "
                             "use SigmaCalculator for 2N x 2N systems")
        key = (id(engine), getattr(engine, "generation", 0), self._version, float(conv),
               tuple(g._version for g in self.gList), int(self.force_iters))
        if key in self._lowered:
            return self._lowered[key][1]
        self._release()
        xi = self.Xi if self.Sdict['sss'] == 0 else None
        h = engine.sigma_bethe(self.indsLists, self.nIndLists, [g.H for g in self.gList],
                               [np.stack(g.Slist) for g in self.gList],
                               [np.stack(g.Vlist) for g in self.gList], xi, self.eta, conv, BETHE_MIX,
                               BETHE_MAX_ITER, self.force_iters)
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

    def _expand_spin(self, sig):
        if self.spin in ('u', 'ro'):
            return np.kron(np.eye(2), sig)
        if self.spin == 'g':
            return np.kron(sig, np.eye(2))
        return sig

    # ---- reference protocol ------------------------------------------------
    def sigma(self, E, i, conv=SURFACE_GREEN_CONVERGENCE):
        eng = self._engine()
        sig = eng.sigma_eval(self._negf_lower(eng, conv), i, [E], self.num_contacts)[0]
        return self._expand_spin(sig)

    def sigmaTot(self, E, conv=SURFACE_GREEN_CONVERGENCE):
        eng = self._engine()
        sig = eng.sigma_eval(self._negf_lower(eng, conv), None, [E], self.num_contacts)[0]
        return self._expand_spin(sig)

    def sigma_batch(self, Elist, i=None, conv=SURFACE_GREEN_CONVERGENCE):
        eng = self._engine()
        out = eng.sigma_eval(self._negf_lower(eng, conv), i, Elist, self.num_contacts)
        return out, eng.last_iters.copy(), eng.last_converged.copy()

    def getSigma(self, Elist=(None, None), conv=SURFACE_GREEN_CONVERGENCE):
        E0 = self.gList[0].fermi if Elist[0] is None else Elist[0]
        E1 = self.gList[-1].fermi if Elist[1] is None else Elist[1]
        return (self.sigma(E0, 0, conv), self.sigma(E1, -1, conv))

    def updateFermi(self, i, Ef):
        self.gList[i].updateH(Ef)

    def setF(self, F, muL, muR):
        self.F = F
        if self.gList[0].fermi != muL:
            self.updateFermi(0, muL)
        if self.gList[-1].fermi != muR:
            self.updateFermi(-1, muR)
