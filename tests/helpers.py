"""Shared input generators for the tests (seeded, synthetic)."""
import numpy as np


def rel_fro(a, b):
    a = np.asarray(a); b = np.asarray(b)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d


def random_system(N, seed):
    """Well-conditioned synthetic F (spectrum ~[-2.8,2.8] eV) and PD overlap S."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, N))
    F = (A + A.T) * (1.0 / np.sqrt(2 * N)) * 2
    B = rng.standard_normal((N, N))
    S = np.eye(N) + 0.1 * (B + B.T) / np.sqrt(2 * N)
    return F, S


def const_sigma_pair(N, S, nc, gamma=0.1):
    """formSigma-style constant contacts: -i*gamma on the first / last nc diagonals."""
    from oracle import form_sigma
    left = list(range(nc)); right = list(range(N - nc, N))
    return [left, right], form_sigma(left, -1j * gamma, N, S), form_sigma(right, -1j * gamma, N, S)


def chain_lead(nc, seed, scale_b=0.2):
    """Lead unit cell of SURVEY.md section 8d config C3."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((nc, nc)); alpha = (a + a.T) * 0.5 * 0.5
    beta = rng.standard_normal((nc, nc)) * scale_b
    s = rng.standard_normal((nc, nc)); Salpha = np.eye(nc) + 0.05 * (s + s.T) * 0.5 / np.sqrt(nc)
    Sbeta = 0.05 * rng.standard_normal((nc, nc)) / np.sqrt(nc)
    return alpha, Salpha, beta, Sbeta


class MockSigma:
    """A foreign provider in the reference's duck-typed protocol (the shape of
    tests/test_computation_consistency.py:23-45): dense, energy dependent, not
    known to the engine -> exercises the host-callback path."""
    def __init__(self, base, contacts):
        self.base = base
        self.contacts = contacts
        self.size = base.shape[0]

    def sigmaTot(self, E):
        return self.base + 1j * (0.01 * E * np.eye(self.size) + 0.001)

    def sigma(self, E, ind):
        if ind >= len(self.contacts):
            ind = 0
        return self.contacts[ind] + 1j * (0.005 * E * np.eye(self.size) + 0.001)
