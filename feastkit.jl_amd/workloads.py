"""Synthetic inputs of the BASELINE.json configurations, closed-form spectra.

Used by ``bench.py`` and the measurement scripts under ``tools/``; the tests build the same
problems through the oracle's generators, so the two stay independent.
"""
from __future__ import annotations

import numpy as np

SEED = 20260515


def laplacian_3d_pencil(nx=50, ny=40, nz=25, shift=0.1):
    """cfg 3/4: A = 7-point Dirichlet Laplacian (x fastest), B = I + shift*A, CSR, sorted indices.

    Returns (A, B, lam) with lam the sorted generalized eigenvalues mu/(1+shift*mu).
    """
    import scipy.sparse as sp

    def t(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")
    Ix, Iy, Iz = sp.identity(nx), sp.identity(ny), sp.identity(nz)
    A = sp.csr_matrix(sp.kron(Iz, sp.kron(Iy, t(nx))) + sp.kron(Iz, sp.kron(t(ny), Ix)) + sp.kron(t(nz), sp.kron(Iy, Ix)))
    A.sort_indices()
    B = sp.csr_matrix(sp.identity(A.shape[0], format="csr") + shift * A)
    B.sort_indices()
    mx = 2 - 2 * np.cos(np.arange(1, nx + 1) * np.pi / (nx + 1))
    my = 2 - 2 * np.cos(np.arange(1, ny + 1) * np.pi / (ny + 1))
    mz = 2 - 2 * np.cos(np.arange(1, nz + 1) * np.pi / (nz + 1))
    mu = np.sort((mx[:, None, None] + my[None, :, None] + mz[None, None, :]).ravel())
    return A, B, np.sort(mu / (1 + shift * mu))


def reflected_diagonal(d, seed=SEED, nreflect=2, complex_reflectors=False):
    """cfg 2: A = H2 H1 diag(d) H1 H2 with seeded unit Householder reflectors; eigenvalues = d."""
    n = d.shape[0]
    rng = np.random.default_rng(seed)
    A = np.diag(d.astype(np.complex128 if complex_reflectors else np.float64))
    for _ in range(nreflect):
        v = rng.standard_normal(n)
        if complex_reflectors:
            v = v + 1j * rng.standard_normal(n)
        v = v / np.linalg.norm(v)
        A = A - 2 * np.outer(A @ v, v.conj())
        A = A - 2 * np.outer(v, v.conj() @ A)
    if not complex_reflectors:
        A = 0.5 * (A + A.T)
    return A


def disc_spectrum_general(N=8192, radius=33.05, coupling=0.05, seed=SEED):
    """cfg 5: A = H2 H1 (diag(delta) + coupling*U) H1 H2, complex reflectors, U strictly upper
    seeded complex Gaussian / sqrt(N), delta uniform in the disc |z| <= radius*sqrt(N/8192).

    Returns (A, delta); the eigenvalues are delta exactly (similarity of a triangular matrix).
    """
    rng = np.random.default_rng(seed)
    rad = radius * np.sqrt(N / 8192.0) * np.sqrt(rng.random(N))
    delta = rad * np.exp(2j * np.pi * rng.random(N))
    U = np.triu(rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N)), 1) / np.sqrt(N)
    A = np.diag(delta) + coupling * U
    del U
    for _ in range(2):
        v = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        v /= np.linalg.norm(v)
        A -= 2 * np.outer(A @ v, v.conj())
        A -= 2 * np.outer(v, v.conj() @ A)
    return A, delta
