"""Matrix ingest for the :hip backend.

* readers for the compact MatrixMarket-like fixtures of the FEAST example ports
  (examples/feast/utils.jl:15-170): first line ``n n nnz``, then ``i j value`` (real) or
  ``i j re im`` (complex), 1-based, no banner.  Dense readers assign (the last duplicate wins,
  utils.jl:21-28); sparse readers sum duplicates like ``sparse(row, col, val, n, n)`` (:67);
  banded readers return LAPACK general band storage with the diagonal in row ``k_upper``
  (0-based; ``k_upper + 1`` in Julia, :110-116);
* ``julia_csc`` / ``HipEngine.set_problem_csc``: the arrays of a ``SparseMatrixCSC{T,Int64}``
  (1-based ``colptr``/``rowval``) handed to ``feasthip_set_csr(storage=CSC, index_base=1)``; the
  library transposes on ingest, so a complex Hermitian or general matrix arrives as ``A`` and
  not as ``A^T``.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def _read_coo(path, complex_values):
    with open(path, "r") as f:
        header = f.readline().split()
        n, nnz = int(header[0]), int(header[2])
        rows = np.empty(nnz, dtype=np.int64)
        cols = np.empty(nnz, dtype=np.int64)
        vals = np.empty(nnz, dtype=np.complex128 if complex_values else np.float64)
        for k in range(nnz):
            parts = f.readline().split()
            if len(parts) < (4 if complex_values else 3):
                raise ValueError(f"{path}: entry {k + 1} of {nnz} is short or missing")
            rows[k], cols[k] = int(parts[0]), int(parts[1])
            vals[k] = complex(float(parts[2]), float(parts[3])) if complex_values else float(parts[2])
    if nnz and (rows.min() < 1 or cols.min() < 1 or rows.max() > n or cols.max() > n):
        raise ValueError(f"{path}: index outside 1..{n}")
    return n, rows - 1, cols - 1, vals


def _dense(path, complex_values):
    n, r, c, v = _read_coo(path, complex_values)
    A = np.zeros((n, n), dtype=v.dtype, order="F")
    for k in range(len(v)):            # assignment, in file order: the last duplicate wins
        A[r[k], c[k]] = v[k]
    return A


def read_mm_dense_real(path):
    return _dense(path, False)


def read_mm_dense_complex(path):
    return _dense(path, True)


def read_mm_sparse_real(path):
    n, r, c, v = _read_coo(path, False)
    return sp.csc_matrix(sp.coo_matrix((v, (r, c)), shape=(n, n)))      # duplicates summed


def read_mm_sparse_complex(path):
    n, r, c, v = _read_coo(path, True)
    return sp.csc_matrix(sp.coo_matrix((v, (r, c)), shape=(n, n)))


def _banded(path, complex_values):
    n, r, c, v = _read_coo(path, complex_values)
    kl = int(max(0, (r - c).max())) if len(v) else 0
    ku = int(max(0, (c - r).max())) if len(v) else 0
    band = np.zeros((kl + ku + 1, n), dtype=v.dtype, order="F")
    for k in range(len(v)):
        band[ku + r[k] - c[k], c[k]] = v[k]
    return band, kl, ku


def read_banded_real(path):
    return _banded(path, False)


def read_banded_complex(path):
    return _banded(path, True)


def banded_to_dense(band, kl, ku):
    """Expand LAPACK general band storage back to a dense matrix (for the dense LU path)."""
    n = band.shape[1]
    A = np.zeros((n, n), dtype=band.dtype, order="F")
    for j in range(n):
        for i in range(max(0, j - ku), min(n, j + kl + 1)):
            A[i, j] = band[ku + i - j, j]
    return A


def write_mm(path, A):
    """Write the compact format (used by the tests and to export synthetic workloads)."""
    M = sp.coo_matrix(A)
    cplx = np.iscomplexobj(M.data)
    with open(path, "w") as f:
        f.write(f"{M.shape[0]} {M.shape[1]} {M.nnz}\n")
        for i, j, v in zip(M.row, M.col, M.data):
            f.write(f"{i + 1} {j + 1} {float(v.real)!r} {float(v.imag)!r}\n" if cplx else f"{i + 1} {j + 1} {float(v)!r}\n")


def julia_csc(A):
    """(colptr, rowval, nzval) of ``SparseMatrixCSC{T,Int64}(A)``: 1-based Int64 arrays, sorted rows."""
    M = sp.csc_matrix(A)
    M.sort_indices()
    return M.indptr.astype(np.int64) + 1, M.indices.astype(np.int64) + 1, np.ascontiguousarray(M.data)


# ---- band storage of the reference's banded drivers (src/banded/feast_banded.jl:1-7, 205-271, 488-509)
def band_upper_to_csr(Ab, k, kind="symmetric"):
    """Upper band storage, (k+1) x N with A(i, j) = Ab[k + i - j, j] for i <= j (0-based), to a full
    CSR matrix; the lower triangle is the mirror image: transposed ("symmetric",
    "complex_symmetric") or conjugate-transposed ("hermitian")."""
    Ab = np.asarray(Ab)
    N = Ab.shape[1]
    if Ab.shape[0] < k + 1:
        raise ValueError("A matrix storage insufficient for k")
    diags, offs = [Ab[k, :]], [0]
    for d in range(1, k + 1):
        up = Ab[k - d, d:]
        diags += [up, np.conj(up) if kind == "hermitian" else up]
        offs += [d, -d]
    return sp.csr_matrix(sp.diags(diags, offs, shape=(N, N)))


def band_general_to_csr(Ab, k):
    """General band storage, (2k+1) x N with A(i, j) = Ab[k + i - j, j], |i - j| <= k."""
    Ab = np.asarray(Ab)
    N = Ab.shape[1]
    diags, offs = [], []
    for d in range(-k, k + 1):          # d = j - i
        row = k - d
        if 0 <= row < Ab.shape[0]:
            diags.append(Ab[row, max(d, 0):N + min(d, 0)])
            offs.append(d)
    return sp.csr_matrix(sp.diags(diags, offs, shape=(N, N)))


def csr_to_band_upper(A, k):
    """Inverse of band_upper_to_csr (upper triangle only)."""
    A = sp.csr_matrix(A)
    N = A.shape[0]
    Ab = np.zeros((k + 1, N), dtype=A.dtype)
    for d in range(k + 1):
        Ab[k - d, d:] = A.diagonal(d)
    return Ab
