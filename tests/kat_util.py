"""Helpers shared by the oracle-pinning tests (CPU) and the GPU parity tests."""
import json
import os

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))


def load_kats():
    with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
        return json.load(f)


def cplx(v):
    return complex(v[0], v[1])


def cmat(M):
    return np.array([[cplx(v) for v in row] for row in M], dtype=np.complex128)


def tridiag(n, dtype=np.float64):
    return (np.diag(2.0 * np.ones(n)) - np.diag(np.ones(n - 1), 1) - np.diag(np.ones(n - 1), -1)).astype(dtype)


def sparse_tridiag(n):
    return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")
