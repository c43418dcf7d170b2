"""GPU parity tests of the RCI job server (feastkit.jl_amd/rci.py::HipRciServer): the reference's
job protocol (src/kernel/feast_kernel.jl) with jobs 10/11/30/40 serviced through the C ABI
(feasthip_shifted_solve / feasthip_matmul), against the oracle's exact-job restatement."""
import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

import feast_oracle as fo
import feastkit_jl_amd as fk
import rci_callers as rci          # the job state machines and caller loops are test infrastructure (tests/rci_callers.py)

pytestmark = pytest.mark.gpu


def tridiag(n):
    return np.diag(2.0 * np.ones(n)) - np.diag(np.ones(n - 1), 1) - np.diag(np.ones(n - 1), -1)


def fpm_with(**kw):
    fpm = fk.feastinit()
    for k, v in kw.items():
        fpm[int(k[1:])] = v
    return fpm


@pytest.mark.parametrize("case", [(None, 0.2, 1.3, 8, 8), (None, 0.1, 0.7, 6, 16), ("B", 0.1, 0.7, 6, 8)])
def test_srci_dense_device_jobs_match_oracle(engine, case):
    which, lo, hi, M0, ne = case
    n = 30
    A = tridiag(n)
    B = None
    if which == "B":
        B = np.diag(1.0 + 0.5 * np.random.default_rng(5).random(n)) + 0.05 * tridiag(n)
    ev = sla.eigh(A, B, eigvals_only=True)
    inside = ev[(ev > lo) & (ev < hi)]
    want = fo.rci_symmetric(A, B, lo, hi, M0, ne=ne, fpm3=11, fpm4=12)
    srv = rci.HipRciServer(engine, A, B, solver="direct")
    got = rci.rci_solve_symmetric(srv, lo, hi, M0, fpm_with(f2=ne, f3=11, f4=12))
    assert (got.info, got.M) == (want.info, want.M) and abs(got.loop - want.loop) <= 1
    assert got.M == len(inside)
    assert np.allclose(got.lambda_, inside, atol=1e-8)
    assert np.allclose(got.lambda_, want.lam, atol=1e-8)
    # one LU per contour node for the whole solve: job 10 is free after the first loop
    assert srv.solves == ne * (got.loop + 1)


def test_srci_factor_cache_counts(engine):
    """feast_dense.jl:458,487-497 keeps one factorisation per shift; so does the device server."""
    A = tridiag(40)
    srv = rci.HipRciServer(engine, A, None, solver="direct")
    nfact = []
    orig = engine.shifted_solve

    def counting(z, dX, m):
        out = orig(z, dX, m)
        nfact.append(engine.last_stats["factorizations"])
        return out
    engine.shifted_solve = counting
    try:
        r = rci.rci_solve_symmetric(srv, 0.2, 1.3, 10, fpm_with(f2=8, f3=10, f4=6))
    finally:
        del engine.shifted_solve
    assert r.info == 0 and r.loop >= 1
    assert sum(nfact) == 8 and sum(nfact[:8]) == 8       # all factorisations happen in the first sweep


def test_srci_sparse_krylov_jobs(engine):
    """CSR input: job 11 runs the batched BiCGStab solver; matrix-free flavour (rhs = work)."""
    def t(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    nx, ny = 14, 11
    A = sp.csr_matrix(sp.kron(sp.identity(ny), t(nx)) + sp.kron(t(ny), sp.identity(nx)))
    ev = np.sort((2 - 2 * np.cos(np.arange(1, nx + 1) * np.pi / (nx + 1)))[:, None]
                 + (2 - 2 * np.cos(np.arange(1, ny + 1) * np.pi / (ny + 1)))[None, :], axis=None)
    lo, hi = 0.0, 0.5 * (ev[4] + ev[5])
    inside = ev[:5]
    srv = rci.HipRciServer(engine, A, None, solver="bicgstab", rtol=1e-13, maxit=5000)
    got = rci.rci_solve_symmetric(srv, lo, hi, 6, fpm_with(f2=8, f3=10, f4=20), matrix_free=True)
    want = fo.rci_symmetric(A, None, lo, hi, 6, ne=8, fpm3=10, fpm4=20, rhs_uses_B=False)
    assert (got.info, got.M) == (want.info, want.M) == (0, len(inside))
    assert abs(got.loop - want.loop) <= 1
    assert np.allclose(got.lambda_, inside, atol=1e-8)
    assert got.epsout <= 1e-10


def test_hrci_device_jobs_match_oracle(engine):
    n = 24
    rng = np.random.default_rng(11)
    H = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = np.diag(np.linspace(0.0, 6.0, n)) + 0.05 * (H + H.conj().T)
    ev = np.linalg.eigvalsh(A)
    lo, hi = 0.5 * (ev[4] + ev[5]), 0.5 * (ev[11] + ev[12])
    want = fo.rci_hermitian(A, None, lo, hi, 8, ne=8, fpm3=11, fpm4=2)
    got = rci.rci_solve_hermitian(rci.HipRciServer(engine, A, None), lo, hi, 8, fpm_with(f2=8, f3=11, f4=2))
    assert (got.info, got.M, got.loop) == (want.info, want.M, want.loop)
    assert np.allclose(got.lambda_, want.lam, atol=1e-8)
    assert np.allclose(got.res, want.res, rtol=1e-3, atol=1e-10)


@pytest.mark.parametrize("generalized", [False, True])
def test_grci_device_jobs_match_oracle(engine, generalized):
    n = 20
    rng = np.random.default_rng(3)
    T = np.diag(np.linspace(-3, 3, n) + 1j * rng.uniform(-1, 1, n)) + 0.1 * np.triu(rng.standard_normal((n, n)), 1)
    S = rng.standard_normal((n, n)) + n * np.eye(n)
    A = S @ T @ np.linalg.inv(S)
    B = np.diag(1.0 + rng.random(n)).astype(complex) if generalized else None
    ev = np.linalg.eigvals(A if B is None else np.linalg.solve(B, A))
    c = 0.3 + 0.1j
    dist = np.sort(np.abs(ev - c))
    r = 0.5 * (dist[6] + dist[7])
    maxloop = 8 if generalized else 20
    want = fo.feast_general(A, B, c, r, 10, ne=16, fpm3=10, fpm4=maxloop)
    got = rci.rci_solve_general(rci.HipRciServer(engine, A, B), c, r, 10, fpm_with(f8=16, f3=10, f4=maxloop))
    assert (got.info, got.M) == (want.info, want.M) == (0, 7)
    assert abs(got.loop - want.loop) <= 1
    inside = ev[np.abs(ev - c) <= r]
    key = lambda x: (round(x.real, 6), round(x.imag, 6))
    assert np.allclose(sorted(got.lambda_, key=key), sorted(inside, key=key), atol=1e-8)
    assert np.allclose(sorted(got.lambda_, key=key), sorted(want.lam, key=key), atol=1e-8)


def test_matfree_fixtures_with_device_linear_solver(engine):
    """test/test_matrix_free.jl:55-185 with linear_solver(Y, z, X) = the device shifted solve."""
    A = np.diag([1.0, 2.0, 3.0, 4.0, 5.0])
    srv = rci.HipRciServer(engine, A, None)
    solve = srv.linear_solver()
    X = np.random.default_rng(0).standard_normal((5, 3))
    Y = np.zeros((5, 3), dtype=complex)
    z = 2.2 + 0.7j
    solve(Y, z, X)
    assert np.allclose((z * np.eye(5) - A) @ Y, X, atol=1e-12)
    r = rci.rci_solve_symmetric(srv, 1.5, 4.5, 5, fk.feastinit(), matrix_free=True)
    assert r.info == 0 and r.M == 3 and np.allclose(np.sort(r.lambda_), [2.0, 3.0, 4.0], atol=1e-10)
    n = 100
    r = rci.rci_solve_symmetric(rci.HipRciServer(engine, tridiag(n), None), 0.8, 1.2, 8, fpm_with(f3=8, f4=20), matrix_free=True)
    ev = 2 - 2 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))
    inside = ev[(ev >= 0.8) & (ev <= 1.2)]
    assert r.info == 0 and r.M == len(inside) and np.allclose(r.lambda_, inside, atol=1e-7)


def test_linear_solver_failure_raises(engine):
    """A singular shifted matrix must surface as an exception (-> info 8 in feast_matfree_srci!)."""
    A = np.diag([1.0, 2.0, 3.0])
    solve = rci.HipRciServer(engine, A, None).linear_solver()
    with pytest.raises(RuntimeError):
        solve(np.zeros((3, 2), complex), 2.0 + 0.0j, np.ones((3, 2)))
