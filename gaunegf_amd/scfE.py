"""
Density step of the energy-dependent NEGF-SCF cycle without Gaussian -- the part of
``gauNEGF/scfE.py`` that drives the energy-grid hot path.

    NEGFE.setVoltage          scf.py:318-370 (chemical potentials; the E-field pushed into the
                              Gaussian interface is not part of this class) + scfE.py:183-205
    NEGFE.setIntegralLimits   scfE.py:207-232
    NEGFE.getHOMOLUMO         scf.py:297-316
    NEGFE.getSigma            scfE.py:283-298
    NEGFE.FockToP             scfE.py:301-462   (contour + real-axis + bias-window integrals,
                                                 Fermi search, level occupations)
    NEGFE.PMix                scf.py:597-661    (damping and Pulay/DIIS mixing of the density matrix)
    NEGFE.SCF                 scf.py:663-800    (the cycle FockToP -> PMix -> new Fock matrix, with a
                                                 caller-supplied Fock model: PToFock calls Gaussian)

The reference keeps F in Hartree and multiplies by ``har_to_eV`` at every use; here the Fock
matrix is handed over (and stored) in eV.  ``ne`` is the electron count the search aims for
(``bar.ne``; halved internally for spin 'r' exactly as scfE.py:374-376).  The 'predict' Fermi
method uses the closed-form ``density()`` / ``bisectFermi()`` (density.py:276-382, host numpy);
every energy-grid integral runs on the GPU engine through ``gaunegf_amd.density``.
"""
import numpy as np
from scipy.linalg import fractional_matrix_power

from ._hostblas import limited_call
from .config import (ADAPTIVE_INTEGRATION_TOL, ENERGY_MIN, FERMI_CALCULATION_TOL, TEMPERATURE, SCF_DAMPING,
                     SCF_CONVERGENCE_TOL, SCF_MAX_CYCLES, PULAY_MIXING_SIZE)
from .density import (bisectFermi, density, calcEmin, calcFermiBisect, calcFermiMuller, calcFermiPolyFit, calcFermiSecant,
                      densityComplex, densityComplexN, densityGrid, densityGridN, densityReal,
                      densityEquilibriumN, densityRealN, integralFit, integralFitNEGF)

har_to_eV = 27.211386   # eV/Hartree (scfE.py:44)


class NEGFE:
    @limited_call
    def __init__(self, F_eV, S, g, ne, spin='r', T=TEMPERATURE, Eminf=ENERGY_MIN, fock_builder=None):
        """``g``: contact object with the reference's protocol (sigma, sigmaTot, setF, F, S);
        ``fock_builder(P) -> F_eV``: optional model replacing Gaussian's PToFock for ``SCF``."""
        self.F = np.array(F_eV)
        self.S = np.array(S)
        self.g = g
        self.ne = ne
        self.spin = spin
        self.T = T
        self.Eminf = Eminf
        self.fock_builder = fock_builder
        self.X = np.array(fractional_matrix_power(self.S, -0.5))            # scf.py:181
        self.fermi = None
        self.updFermi = False
        self.fermiMethod = 'muller'
        self.qV = 0.0
        self.mu1 = self.mu2 = None
        self.convLevel = 9999.0
        self.P = None
        self.P_in = None
        self.nelec = 0.0
        self.setIntegralLimits(Emin=0.0, tol=None)                          # placeholders until setVoltage
        self.tol = ADAPTIVE_INTEGRATION_TOL

    # ------------------------------------------------------------------ set-up
    def nelec_target(self):
        return self.ne / 2 if self.spin == 'r' else self.ne                 # scfE.py:374-376

    def getHOMOLUMO(self):
        orbs = np.sort(np.linalg.eigvals(self.X @ self.F @ self.X))         # scf.py:310-311 (F already in eV)
        n = int(round(self.nelec_target()))
        return np.real(orbs[n - 1:n + 1])

    def setVoltage(self, qV, fermi=np.nan, Emin=None, Eminf=None, fermiMethod='muller'):
        if np.isnan(fermi):                                                 # scf.py:349-357
            self.updFermi = True
            fermi = float(np.sum(self.getHOMOLUMO()) / 2) if self.fermi is None else self.fermi
        else:
            self.updFermi = False
        if Emin is not None:
            self.Emin = Emin
        if Eminf is not None:
            self.Eminf = Eminf
        self.fermi = fermi
        self.qV = qV
        self.mu1 = fermi + qV / 2
        self.mu2 = fermi - qV / 2
        self.g.setF(self.F, self.mu1, self.mu2)                             # scfE.py:200
        if self.mu1 != self.mu2 and self.N1 is not None:
            self.Nnegf = 50                                                 # scfE.py:201-202
        if self.updFermi:
            self.fermiMethod = fermiMethod

    def setIntegralLimits(self, N1=None, N2=None, Nnegf=None, tol=ADAPTIVE_INTEGRATION_TOL, Emin=None):
        if Emin is None and tol is not None:                                # scfE.py:224-227
            self.Emin = calcEmin(self.F, self.S, self.g)
        else:
            self.Emin = Emin
        self.tol = tol
        self.N1 = N1
        self.N2 = N2
        self.Nnegf = Nnegf

    def fitIntegralLimits(self):
        """The limit-fitting half of integralCheck (scfE.py:262-270): Emin, N1, N2 (and Nnegf under bias)."""
        self.Emin, self.N1, self.N2 = integralFit(self.F, self.S, self.g, self.fermi, self.Eminf, self.tol)
        if self.mu1 != self.mu2:
            self.Nnegf = integralFitNEGF(self.F, self.S, self.g, self.fermi, self.qV, self.Eminf, self.tol, self.T)

    def getSigma(self, E):
        return (self.g.sigma(E, 0), self.g.sigma(E, -1))

    # ------------------------------------------------------------------ the density step
    def saveMAT(self, matfile="out.mat"):
        """MATLAB-format results file (scf.py:823-843): keys F (eV), sig1, sig2 (contact self-energies at the Fermi
        level), S, fermi, qV, spin, den, conv -- the file ``transport.currentF`` reads.  Returns the Fock matrix
        in the orthogonalised basis, X F X, as the reference does.  (The reference stores F in Hartree and writes
        F * har_to_eV; here F is kept in eV.)"""
        import scipy.io as io
        sigma1, sigma2 = self.getSigma(self.fermi)
        matdict = {"F": self.F, "sig1": sigma1, "sig2": sigma2, "S": self.S, "fermi": self.fermi, "qV": self.qV,
                   "spin": self.spin, "den": self.P, "conv": self.convLevel}
        io.savemat(matfile, matdict)
        return self.X @ self.F @ self.X

    @limited_call                       # (the step's host eigenproblems and products: _hostblas.py)
    def FockToP(self):
        """Density matrix for the current Fock matrix (scfE.py:301-462).  Returns the sorted
        orbital energies and their occupations; ``self.P`` holds the density matrix."""
        F, S, g = self.F, self.S, self.g
        # fixed grids at a given Fermi level: the real-axis integral and the contour are independent -- one pass of the
        # engine over both grids (density.densityEquilibriumN), the same two sums added in the same order
        P2_fixed = None
        if self.N2 is not None and self.N1 is not None and not self.updFermi:
            P, P2_fixed = densityEquilibriumN(F, S, g, self.Eminf, self.Emin, self.mu1, self.N2, self.N1, self.T)
        elif self.N2 is None:                                               # scfE.py:316-320
            self.Emin = calcEmin(F, S, g)
            P = densityReal(F, S, g, self.Eminf, self.Emin, self.tol, T=0)
        else:
            P = densityRealN(F, S, g, self.Eminf, self.Emin, self.N2, T=0, showText=False)
        nLower = np.trace(S @ P).real

        def compContourP2(mu):                                              # scfE.py:324-328
            if self.N1 is not None:
                return densityComplexN(F, S, g, self.Emin, mu, N=self.N1, T=self.T, showText=False)
            return densityComplex(F, S, g, self.Emin, mu, tol=self.tol, T=self.T)

        if self.updFermi:
            fermi_old = self.fermi + 0.0
            conv = min(self.convLevel, FERMI_CALCULATION_TOL)
            method = self.fermiMethod.lower()
            if method == 'predict':                                         # scfE.py:333-361
                # constant-self-energy estimate of the Fermi shift: the closed-form electron count at the
                # current level, corrected by the electrons the last density matrix was off by
                X = self.X
                sig1, sig2 = self.getSigma(self.fermi)
                Fbar = X @ (F + sig1 + sig2) @ X
                GamBar = X @ ((sig1 - sig1.conj().T) * 1j + (sig2 - sig2.conj().T) * 1j) @ X
                D, V = np.linalg.eig(Fbar)
                Vc = np.linalg.inv(V.conj().T)
                Ncurr = np.trace(density(V, Vc, D, GamBar, self.Eminf, self.fermi)).real
                dN = self.ne - self.updateN()
                if self.spin == 'r':
                    dN /= 2
                dN -= nLower
                Nsearch = Ncurr + dN
                print('CONSTANT SELF-ENERGY APPROXIMATION:')
                if Nsearch > 0 and Nsearch < len(F):
                    self.fermi = bisectFermi(V, Vc, D, GamBar, Ncurr + dN, conv, self.Eminf)
                    print(f'Fermi Energy set to {self.fermi:.2f} eV, shifting by {dN:.2E} electrons ')
                else:
                    print('Warning: Local sigma approximation not valid, Fermi energy not updated...')
                print('Calculating equilibrium density matrix:')
                P = P + compContourP2(self.mu1)
            if method not in ('muller', 'secant', 'bisect', 'poly', 'predict'):
                raise Exception("Error: invalid Fermi search method, needs to be 'muller', 'secant', 'bisect' "
                                "or 'predict' or 'default'")
            methodFail = False
            uBound = lBound = None
            ne = self.nelec_target()
            same_mu = self.mu1 == self.mu2
            if method == 'poly':                                            # scfE.py:371-387
                self.fermi, dE, P2, dN, uBound, lBound = calcFermiPolyFit(g, ne - nLower, self.Emin, fermi_old,
                                                                         self.N1, tol=self.tol, conv=conv, T=self.T)
                methodFail = dN > conv
            elif method == 'muller':                                        # scfE.py:389-405
                self.fermi, dE, P2, dN, uBound, lBound = calcFermiMuller(g, ne - nLower, self.Emin, fermi_old,
                                                                        self.N1, tol=self.tol, conv=conv, T=self.T)
                methodFail = dN > conv
            elif method == 'secant':                                        # scfE.py:407-423
                self.fermi, dE, P2, dN = calcFermiSecant(g, ne - nLower, self.Emin, fermi_old, self.N1,
                                                         tol=self.tol, conv=conv, T=self.T)
                methodFail = dN > conv
            if method not in ('bisect', 'predict'):
                if methodFail:
                    print(f'Switching to BISECT method (Fermi error = {dE:.2E} eV)')
                    fermi_old = self.fermi + 0.0
                else:
                    print(f'Fermi Energy set to {self.fermi:.2f} eV, error = {dE:.2E} eV ')
                    P = P + P2 if same_mu else P + compContourP2(self.mu1)
            if method == 'bisect' or (methodFail and method != 'predict'):  # scfE.py:425-435
                self.fermi, dE, P2 = calcFermiBisect(g, ne - nLower, self.Emin, fermi_old, self.N1, tol=self.tol,
                                                     conv=conv, T=self.T, uBound=uBound, lBound=lBound)
                print(f'Fermi Energy set to {self.fermi:.2f} eV, error = {dE:.2E} eV ')
                P = P + P2 if same_mu else P + compContourP2(self.mu1)
            # shift Emin, mu1, mu2 and refresh the contact self-energies (scfE.py:440-443)
            self.setVoltage(self.qV, fermiMethod=self.fermiMethod)
            self.Emin += self.fermi - fermi_old
            self.g.setF(F, self.mu1, self.mu2)
        else:
            P = P + (P2_fixed if P2_fixed is not None else compContourP2(self.mu1))     # scfE.py:444-446

        if self.mu1 != self.mu2:                                            # scfE.py:449-457
            if self.Nnegf is not None:
                P = P + densityGridN(F, S, g, self.mu1, self.mu2, ind=-1, N=self.Nnegf, T=self.T, showText=False)
            else:
                P = P + densityGrid(F, S, g, self.mu1, self.mu2, ind=-1, tol=self.tol, T=self.T)

        # level occupations in the Lowdin basis (scfE.py:460-468)
        D, V = np.linalg.eigh(self.X @ F @ self.X)
        if getattr(self, "_Xi_of", None) is not self.X:                     # inv(X) of the step before, while X is the same array
            self._Xi, self._Xi_of = np.linalg.inv(self.X), self.X
        Xi = self._Xi
        pshift = V.conj().T @ (Xi @ P @ Xi) @ V
        self.P = np.array(P)
        occList = np.diag(np.real(pshift))
        EList = np.array(np.real(D)).flatten()
        inds = np.argsort(EList)
        return EList[inds], occList[inds]

    def updateN(self):
        """Electron count of the current density matrix, doubled for restricted spin (scf.py:248-266).
        Before the first density step the target count stands in (the reference starts from Gaussian's
        initial density, which holds ``bar.ne`` electrons)."""
        if self.P is None:
            self.nelec = float(self.ne)
        else:
            nOcc = np.real(np.trace(self.P @ self.S))
            self.nelec = 2 * nOcc if self.spin == 'r' else nOcc
        return self.nelec

    # ------------------------------------------------------------------ density mixing (scf.py:597-661)
    def _init_pulay(self, nPulay):
        """History of the last nPulay mixed densities and residuals and the DIIS system (scf.py:191-196):
        B_ij = <dP_i, dP_j>, bordered by -1 with a zero corner; right-hand side (0, ..., 0, -1)."""
        n = len(self.F)
        P0 = np.zeros((n, n), dtype=complex) if self.P_in is None else self.P_in
        self.pList = np.array([P0 for _ in range(nPulay)], dtype=complex)
        self.DPList = np.ones((nPulay, n, n), dtype=complex) * 1e4
        self.pMat = np.ones((nPulay + 1, nPulay + 1), dtype=complex) * -1
        self.pMat[-1, -1] = 0
        self.pB = np.zeros(nPulay + 1)
        self.pB[-1] = -1

    def PMix(self, damping, Pulay=False):
        """Mix the density matrix FockToP just produced (``self.P``) with the one the Fock matrix was built
        from (``self.P_in``; Gaussian's stored density in the reference): damped step, or the DIIS
        combination of the stored history (scf.py:597-661).  Returns (RMSDP, MaxDP) of the diagonal."""
        Pback = self.P_in
        Dense_diff = abs(np.diag(self.P) - np.diag(Pback))
        self.pList[1:, :, :] = self.pList[:-1, :, :]
        self.pList[0, :, :] = Pback + damping * (self.P - Pback)
        self.DPList[1:, :, :] = self.DPList[:-1, :, :]
        self.DPList[0, :, :] = self.P - Pback
        for i, v1 in enumerate(self.DPList):
            for j, v2 in enumerate(self.DPList):
                self.pMat[i, j] = np.sum(v1 * v2)
        if Pulay:
            coeff = np.linalg.solve(self.pMat, self.pB)[:-1]
            print("Applying Pulay Coeff: ", coeff)
            self.P = sum([self.pList[i, :, :] * coeff[i] for i in range(len(coeff))])
            self.pList[0, :, :] = self.P
        else:
            print("Applying Damping value=", damping)
            self.P = self.pList[0, :, :].copy()
        self.P_in = self.P                         # "storeDen": the next Fock matrix is built from this one
        self.updateN()
        print(f'Total number of electrons (NEGF): {self.nelec:.2f}')
        self.MaxDP = max(Dense_diff)
        RMSDP = np.sqrt(np.mean(Dense_diff ** 2))
        print(f'MaxDP: {self.MaxDP:.2E} | RMSDP: {RMSDP:.2E}')
        return RMSDP, self.MaxDP

    # ------------------------------------------------------------------ a model SCF loop
    def SCF(self, conv=SCF_CONVERGENCE_TOL, damping=SCF_DAMPING, maxcycles=SCF_MAX_CYCLES, pulay=True,
            nPulay=PULAY_MIXING_SIZE, P0=None):
        """The NEGF-SCF cycle of scf.py:663-800 with a caller-supplied Fock model in place of Gaussian
        (``fock_builder(P) -> F`` in eV): FockToP, PMix -- damping, and a Pulay/DIIS step every
        (nPulay + 1)-th cycle -- then the new Fock matrix; converged when RMSDP and MaxDP of the diagonal
        fall below ``conv`` (the reference also asks that of Gaussian's energy change).  Returns the list of
        convergence levels, one per cycle."""
        if self.fock_builder is None:
            raise RuntimeError("SCF needs fock_builder(P) -> F in eV (Gaussian is not available here)")
        n = len(self.F)
        self.P_in = np.zeros((n, n), dtype=complex) if P0 is None else np.array(P0, dtype=complex)
        self._init_pulay(nPulay)
        history = []
        Niter = 0
        while True:
            isPulay = bool(pulay) and ((Niter + 1) % (len(self.pList) + 1) == 0)
            self.FockToP()
            RMSDP, MaxDP = self.PMix(damping, isPulay)
            self.F = np.array(self.fock_builder(self.P))
            self.g.setF(self.F, self.mu1, self.mu2)
            self.convLevel = float(max(RMSDP, MaxDP))
            history.append(self.convLevel)
            if self.convLevel < conv or Niter >= maxcycles:
                break
            Niter += 1
        return history
