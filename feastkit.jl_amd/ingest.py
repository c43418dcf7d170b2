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
