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


def _weighted_grid_laplacian(dims, weights):
    """sum over grid edges (i, j) of w_ij (e_i - e_j)(e_i - e_j)^T plus Dirichlet boundary terms: the finite-volume
    diffusion operator -div(k grad u) on a box grid (x fastest) with one conductivity per edge; symmetric positive
    definite, same pattern as the constant-coefficient stencil.  weights[d] has the grid's shape with dims[d] + 1 along
    axis d (edge to the lower neighbour / the boundary on either end)."""
    import scipy.sparse as sp
    n = int(np.prod(dims))
    idx = np.arange(n).reshape(dims[::-1])             # idx[z, y, x] (or [y, x]): x fastest
    diag = np.zeros(n)
    rows, cols, vals = [], [], []
    nd = len(dims)
    for d in range(nd):
        ax = nd - 1 - d                                # numpy axis of grid direction d
        w = weights[d]
        lo = np.take(w, range(0, dims[d]), axis=ax)    # edge below each cell
        hi = np.take(w, range(1, dims[d] + 1), axis=ax)
        diag += (lo + hi).ravel()
        a = np.take(idx, range(0, dims[d] - 1), axis=ax).ravel()
        b = np.take(idx, range(1, dims[d]), axis=ax).ravel()
        wi = np.take(w, range(1, dims[d]), axis=ax).ravel()
        rows += [a, b]
        cols += [b, a]
        vals += [-wi, -wi]
    A = sp.coo_matrix((np.concatenate(vals + [diag]), (np.concatenate(rows + [np.arange(n)]), np.concatenate(cols + [np.arange(n)]))),
                      shape=(n, n)).tocsr()
    A.sort_indices()
    return A


def variable_coefficient_pencil(dims=(50, 40, 25), kind="diag_mass", seed=SEED, contrast=4.0):
    """Pencils whose A and B do NOT commute (no closed-form spectrum; the tests take scipy's shift-invert Lanczos as
    the CPU answer).  A = -div(k grad) on the box grid `dims` with a smooth-times-random conductivity per edge
    (ratio up to `contrast`), Dirichlet boundary.
      kind "diag_mass":  B = diag(rho), a random lumped mass matrix, rho in [1, contrast]
      kind "stiff_mass": B = I + 0.1 * (a second, independently weighted diffusion operator): same pattern as A, SPD
      kind "identity":   B = None (standard problem)
    Returns (A, B)."""
    import scipy.sparse as sp
    rng = np.random.default_rng([seed, len(dims), int(np.prod(dims)), {"diag_mass": 1, "stiff_mass": 2, "identity": 3}[kind]])
    shape = tuple(dims[::-1])

    def edge_weights():
        ws = []
        for d in range(len(dims)):
            s = list(shape)
            s[len(dims) - 1 - d] += 1
            grids = np.meshgrid(*[np.linspace(0.0, 1.0, m) for m in s], indexing="ij")
            smooth = 1.0 + 0.5 * sum(np.sin(2.0 * np.pi * (k + 1) * g + 0.7 * k) for k, g in enumerate(grids)) / len(grids)
            ws.append(smooth * (1.0 + (contrast - 1.0) * rng.random(s)) / contrast * 2.0)
        return ws
    A = _weighted_grid_laplacian(dims, edge_weights())
    n = A.shape[0]
    if kind == "diag_mass":
        B = sp.diags(1.0 + (contrast - 1.0) * rng.random(n)).tocsr()
    elif kind == "stiff_mass":
        B = sp.csr_matrix(sp.identity(n, format="csr") + 0.1 * _weighted_grid_laplacian(dims, edge_weights()))
        B.sort_indices()
    else:
        B = None
    return A, B
