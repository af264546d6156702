"""
Rate-matrix helpers for the eigen-decomposed models (reference: pastml/models/generator.py).

The decomposition itself stays on the host, as in the reference (numpy/LAPACK, once per frequency change,
``CustomRatesModel.py:52,68``); P(t) = A diag(exp(d t)) A^-1 for every branch is computed by the HIP library.
"""
import numpy as np


def save_matrix(states, matrix, outfile):
    """Space-separated matrix with a '# state names' header line (generator.py:4-13)."""
    np.savetxt(outfile, matrix, delimiter=' ', fmt='%.18e', header=' '.join(states))


def get_normalised_generator(frequencies, rate_matrix=None):
    """
    Q = (R o pi) with rows summing to zero, scaled so that the expected rate -sum_i pi_i q_ii is one
    (generator.py:33-51).  ``rate_matrix`` defaults to all-equal rates.
    """
    n = len(frequencies)
    if rate_matrix is None:
        rate_matrix = np.ones(shape=(n, n), dtype=np.float64) - np.eye(n)
    q = rate_matrix * frequencies
    q -= np.diag(q.sum(axis=1))
    q /= -q.diagonal().dot(frequencies)
    return q


def get_diagonalisation(frequencies, rate_matrix=None):
    """
    (d, A, A^-1) with A diag(d) A^-1 = Q, through the same numpy calls as the reference (generator.py:16-30) so that
    the three arrays are bit-identical to the reference's for identical inputs.  Should LAPACK return a complex
    pair for a (numerically) degenerate spectrum, the equivalent symmetric route
    S = Pi^1/2 Q Pi^-1/2 = U L U^T, A = Pi^-1/2 U is used instead (Q is reversible, so its spectrum is real).
    """
    q = get_normalised_generator(frequencies, rate_matrix)
    d, a = np.linalg.eig(q)
    if np.iscomplexobj(d):
        sq = np.sqrt(np.asarray(frequencies, dtype=np.float64))
        s = (q * sq[:, None]) / sq[None, :]
        s = (s + s.T) / 2
        d, u = np.linalg.eigh(s)
        return d, u / sq[:, None], u.T * sq[None, :]
    return d, a, np.linalg.inv(a)


def get_pij_matrix(t, diag, A, A_inv):
    """Host-side numpy form of A diag(exp(d t)) A^-1 (generator.py:54-65); kept for API completeness only."""
    return A.dot(np.diag(np.exp(diag * t))).dot(A_inv)
