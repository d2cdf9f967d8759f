"""
Host-side helper of the constant self-energy provider.  Only ``formSigma`` of
gauNEGF/matTools.py is in scope (the rest of that module is Gaussian I/O).
"""
import numpy as np


def formSigma(inds, V, nsto, S=0):
    """Self-energy matrix of one contact (gauNEGF/matTools.py:39-74).

    ``-1j*1e-9*S`` on every orbital (S = identity when not given), then ``V`` on the
    contact: a scalar goes on the diagonal entries ``inds``, a matrix fills the
    ``ix_(inds, inds)`` block.  O(N^2) setup work, done once on the host."""
    overlap = np.eye(nsto) if isinstance(S, int) else S
    sigma = np.array(-1j * 1e-9 * overlap, dtype=complex)
    if isinstance(V, (int, complex, float)):
        idx = np.asarray(list(inds), dtype=int)
        sigma[idx, idx] = V
    else:
        sigma[np.ix_(inds, inds)] = V
    return sigma
