"""C-ABI contract on the GPU: Julia-style CSC / 1-based ingest, host-pointer entry points,
and the reference's error codes for bad arguments (src/core/feast_types.jl:257-268)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import feastkit_jl_amd as fk

pytestmark = pytest.mark.gpu


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture()
def raw():
    lib = fk.load_library()
    h = C.c_void_p()
    assert lib.feasthip_create(C.byref(h), 0) == 0
    yield lib, h
    lib.feasthip_destroy(h)


def test_csc_one_based_general_complex_matrix(raw):
    """SparseMatrixCSC{ComplexF64,Int64} as Julia hands it over: 1-based colptr/rowval.  The result
    must be for the matrix as the caller defines it (no silent transpose/conjugate)."""
    lib, h = raw
    N, m = 300, 5
    rng = np.random.default_rng(1)
    A = sp.random(N, N, density=0.03, random_state=1, format="csc") + 1j * sp.random(N, N, density=0.03, random_state=2, format="csc")
    A = sp.csc_matrix(A + sp.diags(np.arange(1, N + 1) * (1 + 0.5j)))
    B = sp.csc_matrix(sp.random(N, N, density=0.02, random_state=3, format="csc") * (1 - 0.3j) + sp.identity(N) * 2)
    A.sort_indices(); B.sort_indices()
    pa, ia, va = (A.indptr + 1).astype(np.int64), (A.indices + 1).astype(np.int64), A.data.astype(np.complex128)
    pb, ib, vb = (B.indptr + 1).astype(np.int64), (B.indices + 1).astype(np.int64), B.data.astype(np.complex128)
    rc = lib.feasthip_set_csr(h, N, 1, 1, 1, len(va), _ptr(pa), _ptr(ia), _ptr(va), len(vb), _ptr(pb), _ptr(ib), _ptr(vb))
    assert rc == 0
    X = np.asfortranarray(rng.standard_normal((N, m)) + 1j * rng.standard_normal((N, m)))
    Y = np.zeros_like(X)
    assert lib.feasthip_matmul(h, 0, m, _ptr(X), _ptr(Y)) == 0
    assert np.abs(Y - A @ X).max() <= 1e-12 * np.abs(A @ X).max()
    assert lib.feasthip_matmul(h, 1, m, _ptr(X), _ptr(Y)) == 0
    assert np.abs(Y - B @ X).max() <= 1e-12 * np.abs(B @ X).max()
    # unsorted rows and duplicate entries are accepted (duplicates are summed)
    ia2 = ia.copy(); va2 = va.copy()
    s, e = pa[0] - 1, pa[1] - 1
    ia2[s:e] = ia2[s:e][::-1]; va2[s:e] = va2[s:e][::-1]
    assert lib.feasthip_set_csr(h, N, 1, 1, 1, len(va2), _ptr(pa), _ptr(ia2), _ptr(va2), 0, None, None, None) == 0
    assert lib.feasthip_matmul(h, 0, m, _ptr(X), _ptr(Y)) == 0
    assert np.abs(Y - A @ X).max() <= 1e-12 * np.abs(A @ X).max()
    assert lib.feasthip_matmul(h, 1, m, _ptr(X), _ptr(Y)) == 0          # B = identity now
    assert np.abs(Y - X).max() == 0.0


def test_host_pointer_contour_apply_and_helpers(raw):
    """The host-pointer entry points a ccall shim uses (copy in / copy out, nothing retained)."""
    lib, h = raw
    N, m = 60, 6
    rng = np.random.default_rng(2)
    A = rng.standard_normal((N, N)); A = np.asfortranarray(A + A.T)
    assert lib.feasthip_set_dense(h, N, 0, _ptr(A), N, None, N) == 0
    fpm = fk.feastdefault(fk.feastinit())
    Z, W = fk.feast_contour(-1.0, 1.0, fpm)
    assert lib.feasthip_set_contour(h, len(Z), _ptr(Z), _ptr(W), 2.0) == 0
    assert lib.feasthip_set_solver(h, 0, 1e-12, 0.0, 100, 30, 64, 1) == 0
    Q = np.asfortranarray(rng.standard_normal((N, m)).astype(np.complex128))
    P = np.zeros_like(Q)
    status = np.zeros(8, dtype=np.int32)
    st = fk._lib.FeastHipStats()
    assert lib.feasthip_contour_apply(h, m, _ptr(Q), None, _ptr(P), None, None, _ptr(status), C.byref(st)) == 0
    ref = sum(2 * w * np.linalg.solve(z * np.eye(N) - A, Q) for z, w in zip(Z, W))
    assert np.abs(P - ref).max() <= 1e-10 * np.abs(ref).max() and st.factorizations == 8 and (status == 0).all()
    rank = C.c_int(0)
    assert lib.feasthip_orthonormalize(h, m, _ptr(P), float(np.sqrt(np.finfo(float).eps)), C.byref(rank)) == 0
    assert rank.value == m and np.abs(P.conj().T @ P - np.eye(m)).max() < 1e-12
    Aq = np.zeros((m, m), dtype=np.complex128, order="F"); Bq = np.zeros_like(Aq)
    assert lib.feasthip_project(h, m, _ptr(P), 0, 1, _ptr(Aq), _ptr(Bq)) == 0
    assert np.abs(Aq - P.conj().T @ A @ P).max() < 1e-11 and np.abs(Bq - np.eye(m)).max() == 0
    lam, V = np.linalg.eigh(Aq)
    X = np.zeros_like(P); res = np.zeros(m)
    lamc = lam.astype(np.complex128)
    assert lib.feasthip_ritz_residual(h, m, _ptr(P), _ptr(np.asfortranarray(V)), _ptr(lamc), m, 1, 1, _ptr(X), _ptr(res)) == 0
    Xr = P @ V; Xr /= np.linalg.norm(Xr, axis=0)
    assert np.abs(X - Xr).max() < 1e-11
    assert np.allclose(res, np.linalg.norm(A @ Xr - Xr * lam, axis=0) / np.maximum(abs(lam), 1), atol=1e-12)
    Y = np.zeros_like(Q)
    assert lib.feasthip_shifted_solve(h, 0.3, 0.4, m, _ptr(Q), _ptr(Y), None) == 0
    assert np.abs((0.3 + 0.4j) * Y - A @ Y - Q).max() < 1e-10


def test_error_codes(raw):
    lib, h = raw
    one = np.ones(4)
    # nothing set yet
    Q = np.zeros((4, 1), dtype=np.complex128, order="F")
    assert lib.feasthip_matmul(h, 0, 1, _ptr(Q), _ptr(Q)) == 1                       # Feast_ERROR_N: no matrix
    assert lib.feasthip_set_dense(h, 0, 0, _ptr(one), 1, None, 1) == 1              # N <= 0
    assert lib.feasthip_set_dense(h, 4, 0, None, 4, None, 4) == 1                   # null A
    assert lib.feasthip_set_dense(h, 4, 0, _ptr(np.eye(4, order="F")), 2, None, 4) == 1   # lda < N
    A = np.asfortranarray(np.diag([1.0, 2.0, 3.0, 4.0]))
    assert lib.feasthip_set_dense(h, 4, 0, _ptr(A), 4, None, 4) == 0
    assert lib.feasthip_matmul(h, 0, 0, _ptr(Q), _ptr(Q)) == 2                       # Feast_ERROR_M0: m <= 0
    assert lib.feasthip_matmul(h, 0, 5, _ptr(Q), _ptr(Q)) == 2                       # m > N
    assert lib.feasthip_matmul(h, 2, 1, _ptr(Q), _ptr(Q)) == 7                       # bad operator selector
    P = np.zeros_like(Q)
    assert lib.feasthip_contour_apply(h, 1, _ptr(Q), None, _ptr(P), None, None, None, None) == 9   # no contour: Feast_ERROR_FPM
    assert lib.feasthip_set_contour(h, 0, None, None, 2.0) == 9
    z = np.array([0.5 + 0.5j]); w = np.array([0.1 + 0j])
    assert lib.feasthip_set_contour(h, 1, _ptr(z), _ptr(w), 2.0) == 0
    assert lib.feasthip_set_node_range(h, 0, 2) == 9 and lib.feasthip_set_node_range(h, 1, 0) == 0
    bad = np.array([3], dtype=np.int32)
    assert lib.feasthip_set_node_list(h, 1, _ptr(bad)) == 9
    assert lib.feasthip_set_solver(h, 7, 1e-12, 0.0, 10, 30, 64, 1) == 9             # unknown solver kind
    assert lib.feasthip_set_solver(h, 1, 1e-12, 0.0, 0, 30, 64, 1) == 9              # maxit <= 0
    assert lib.feasthip_set_solver(h, 1, 1e-12, 0.0, 10, 30, 16, 1) == 9             # precision not 32/64
    # malformed CSR: column index out of range
    ptr = np.array([0, 1, 2], dtype=np.int64); idx = np.array([0, 5], dtype=np.int64); val = np.array([1.0, 2.0])
    assert lib.feasthip_set_csr(h, 2, 0, 0, 0, 2, _ptr(ptr), _ptr(idx), _ptr(val), 0, None, None, None) == 1
    assert b"out of range" in lib.feasthip_last_error(h)
    # singular shift with the LU solver -> per-node status 8 (Feast_ERROR_LAPACK), call itself succeeds
    assert lib.feasthip_set_dense(h, 4, 0, _ptr(A), 4, None, 4) == 0
    zs = np.array([2.0 + 0j]); ws = np.array([1.0 + 0j])
    assert lib.feasthip_set_contour(h, 1, _ptr(zs), _ptr(ws), 1.0) == 0
    assert lib.feasthip_set_solver(h, 0, 1e-12, 0.0, 10, 30, 64, 1) == 0
    status = np.zeros(1, dtype=np.int32)
    Q1 = np.asfortranarray(np.ones((4, 1), dtype=np.complex128))
    assert lib.feasthip_contour_apply(h, 1, _ptr(Q1), None, _ptr(P), None, None, _ptr(status), None) == 0
    assert status[0] == 8


def test_column_mask_keeps_initial_guess(engine):
    """feasthip_set_column_mask: masked columns come back as their initial guess summed over the
    nodes -- sum_e w_e q_c / (z_e - lambda_c) with the Ritz warm start -- unmasked columns are solved."""
    import feast_oracle as fo
    import feastkit_jl_amd as fk
    A, B, lam = fo.cfg3_problem(8, 7, 6)
    N = A.shape[0]
    engine.set_problem(A, B)
    fpm = fk.feastdefault(fk.feastinit()); fpm[2] = 4
    Z, W = fk.feast_contour(0.0, 0.9, fpm)
    engine.set_contour(Z, W, 2.0)
    engine.set_real_projection(False)
    engine.set_solver("cocg", rtol=1e-12, atol=0.0, maxit=4000)
    Q = fk.seeded_subspace(N, 4)
    ritz = np.array([0.3, 0.5, 1.4, 2.0])
    try:
        engine.set_column_mask([1, 0, 1, 0])
        dP, status, st = engine.contour_apply(engine.upload(Q), 4, ritz)
    finally:
        engine.set_column_mask(None)
    P = engine.download(dP, 4)
    Sd = [Z[e] * B.toarray() - A.toarray() for e in range(4)]
    for c in range(4):
        if c in (1, 3):
            want = sum(2 * W[e] / (Z[e] - ritz[c]) for e in range(4)) * Q[:, c]
        else:
            want = sum(2 * W[e] * np.linalg.solve(Sd[e], B @ Q[:, c]) for e in range(4))
        assert np.allclose(P[:, c], want, atol=1e-9), c
    lib = engine.lib
    assert lib.feasthip_set_column_mask(engine.h, -1, None) == 2


@pytest.mark.parametrize("solver", ["cocg", "bicgstab", "gmres"])
def test_non_finite_residual_is_never_reported_converged(engine, solver):
    """A NaN in one right-hand-side column: that column's residual norm is not finite.  `!(NaN > target)` is true, so an
    unguarded stop test marks it converged (k_fin_rho tested the target before isfinite; the host gather
    dropped NaN ratios).  The solve must come back with the no-convergence code, the clean columns solved; and the one-shot
    column mask must not leak into this call (same review)."""
    import feast_oracle as fo
    A, B, lam = fo.cfg3_problem(8, 7, 6)
    N = A.shape[0]
    engine.set_problem(A, B)
    engine.set_solver(solver, rtol=1e-10, atol=0.0, maxit=3000, restart=40)
    engine.set_column_mask([0, 0, 0, 0])             # consumed by contour_apply only: must be ignored here
    X = np.asfortranarray(fk.seeded_subspace(N, 4).copy())
    X[3, 2] = np.nan
    z = 0.45 + 0.3j
    dY, rc = engine.shifted_solve(z, engine.upload(X), 4)
    engine.set_column_mask(None)
    assert rc == 5, rc
    Y = engine.download(dY, 4)
    S = z * B.toarray() - A.toarray()
    for c in (0, 1, 3):
        ref = np.linalg.solve(S, X[:, c])
        assert np.abs(Y[:, c] - ref).max() <= 1e-7 * np.abs(ref).max(), (solver, c)
