"""CPU tests of the RCI job protocol (tests/rci_callers.py, the caller side the reference keeps in Julia): the job protocol of
feast_srci!/hrci!/grci! (src/kernel/feast_kernel.jl) driven with a test-only exact job server,
against the oracle's straight-line restatement and the reference's own RCI fixtures."""
import numpy as np
import pytest
import scipy.sparse as sp

import feast_oracle as fo
import feastkit_jl_amd as fk
import rci_callers as rci          # the job state machines and caller loops are test infrastructure (tests/rci_callers.py)
from rci_numpy_server import NumpyRciServer


def tridiag(n):
    return np.diag(2.0 * np.ones(n)) - np.diag(np.ones(n - 1), 1) - np.diag(np.ones(n - 1), -1)


def fpm_with(**kw):
    fpm = fk.feastinit()
    for k, v in kw.items():
        fpm[int(k[1:])] = v
    return fpm


def test_entry_points_initialise_like_reference():
    """test/runtests.jl:78-118: after the init call info == 0 and the next job is FACTORIZE."""
    n, m0 = 6, 3
    for kind in ("s", "h", "g"):
        refs, state = rci.RciRefs(), rci.RciState()
        fpm = fk.feastinit()
        fpm[1] = 0
        work = np.empty((n, m0))
        workc = np.empty((n, m0), dtype=complex)
        lam = np.empty(m0, dtype=complex if kind == "g" else float)
        res = np.empty(m0)
        if kind == "s":
            rci.feast_srci(refs, n, work, workc, np.empty((m0, m0)), np.empty((m0, m0)), fpm, 0.0, 2.0, m0, lam,
                           np.empty((n, m0)), res, state)
        elif kind == "h":
            rci.feast_hrci(refs, n, work, workc, np.empty((m0, m0), complex), np.empty((m0, m0), complex), fpm, 0.0, 2.0,
                           m0, lam, np.empty((n, m0), complex), res, state)
        else:
            rci.feast_grci(refs, n, work, workc, np.empty((m0, m0), complex), np.empty((m0, m0), complex), fpm,
                           1.0 + 0.0j, 2.0, m0, lam, np.empty((n, m0), complex), res, state)
        assert refs.info == 0
        assert refs.ijob == int(fk.FeastRCIJob.Feast_RCI_FACTORIZE)
        assert refs.Ze == state.Zne[0]
        assert fpm[2] == 8 and fpm[8] == 16          # feastdefault! ran


@pytest.mark.parametrize("args,code", [((0, 2, 0.0, 1.0), 1), ((5, 0, 0.0, 1.0), 2), ((5, 6, 0.0, 1.0), 2), ((5, 2, 1.0, 1.0), 3)])
def test_srci_input_errors(args, code):
    N, M0, Emin, Emax = args
    refs, state = rci.RciRefs(), rci.RciState()
    n = max(N, 1)
    m = max(M0, 1)
    rci.feast_srci(refs, N, np.zeros((n, m)), np.zeros((n, m), complex), np.zeros((m, m)), np.zeros((m, m)), fk.feastinit(),
                   Emin, Emax, M0, np.zeros(m), np.zeros((n, m)), np.zeros(m), state)
    assert refs.info == code and refs.ijob == -1


def test_grci_radius_error_and_invalid_job():
    refs, state = rci.RciRefs(), rci.RciState()
    rci.feast_grci(refs, 4, np.zeros((4, 2)), np.zeros((4, 2), complex), np.zeros((2, 2), complex), np.zeros((2, 2), complex),
                   fk.feastinit(), 0j, 0.0, 2, np.zeros(2, complex), np.zeros((4, 2), complex), np.zeros(2), state)
    assert refs.info == 4
    refs.ijob = 77
    with pytest.raises(ValueError):
        rci.feast_grci(refs, 4, np.zeros((4, 2)), np.zeros((4, 2), complex), np.zeros((2, 2), complex), np.zeros((2, 2), complex),
                       fk.feastinit(), 0j, 1.0, 2, np.zeros(2, complex), np.zeros((4, 2), complex), np.zeros(2), state)


def test_srci_job_sequence():
    """One refinement loop issues (10, 11) per node, then 30; the next loop starts at 10 again."""
    A = tridiag(12)
    srv = NumpyRciServer(A)
    r = rci.rci_solve_symmetric(srv, 0.1, 1.1, 6, fpm_with(f2=4, f4=3, f3=10))
    assert r.info == 0
    per_loop = [10, 11] * 4 + [30]
    assert srv.jobs[:9] == per_loop
    assert len(srv.jobs) == 9 * (r.loop + 1)
    assert srv.jobs == per_loop * (r.loop + 1)
    assert len(srv.factors) == 4                       # one factorisation per node, reused every loop


@pytest.mark.parametrize("case", [(None, 0.2, 1.3, 8, 8), (None, 0.1, 0.7, 6, 16), ("B", 0.1, 0.7, 6, 8)])
def test_srci_matches_oracle(case):
    """Configurations the reference's moment variant handles: the RCI kernels have no rank
    compression, so a subspace much larger than the eigenvalue count makes the reduced pencil
    (Sq, Aq) numerically singular after a few loops and the un-normalised residual meaningless --
    in the reference too (cf. src/core/feast_backend_utils.jl:115).  With B != I the kernel's
    residual ||A q - lambda q|| never converges (feast_kernel.jl:244-252 ignores B): the solve runs
    fpm[4] loops and still returns info 0 (:258), eigenvalues correct."""
    import scipy.linalg as sla
    which, lo, hi, M0, ne = case
    n = 30
    A = tridiag(n)
    rng = np.random.default_rng(5)
    B = None
    if which == "B":
        B = np.diag(1.0 + 0.5 * rng.random(n)) + 0.05 * tridiag(n)
    ev = sla.eigh(A, B, eigvals_only=True)
    inside = ev[(ev > lo) & (ev < hi)]
    want = fo.rci_symmetric(A, B, lo, hi, M0, ne=ne, fpm3=11, fpm4=12)
    got = rci.rci_solve_symmetric(NumpyRciServer(A, B), lo, hi, M0, fpm_with(f2=ne, f3=11, f4=12))
    assert (got.info, got.M, got.loop) == (want.info, want.M, want.loop)
    assert got.info == 0 and got.M == len(inside)
    assert np.allclose(got.lambda_, want.lam, atol=1e-9)
    assert np.allclose(got.lambda_, inside, atol=1e-8)
    if B is None:
        assert got.epsout <= 1e-11 and abs(np.log10(got.epsout / want.epsout)) < 1.0
    else:
        assert got.loop == 12 and got.epsout > 1e-3


def test_hrci_matches_oracle_complex_hermitian():
    """Parity with the straight-line restatement on a genuinely complex Hermitian matrix.  The
    kernel sums 2*w_e*Y_e over the upper half contour only and never adds the conjugate half, which
    for complex A is not the spectral projector (same finding as variant A): it does
    not converge, in the reference either -- so this checks the state machine, loop for loop."""
    n = 24
    rng = np.random.default_rng(11)
    H = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = np.diag(np.linspace(0.0, 6.0, n)) + 0.05 * (H + H.conj().T)
    ev = np.linalg.eigvalsh(A)
    lo, hi = 0.5 * (ev[4] + ev[5]), 0.5 * (ev[11] + ev[12])
    for loops in (1, 2, 3):      # fpm[4] <= 0 is reset to the default 20 by feastdefault! (feast_parameters.jl:130)
        want = fo.rci_hermitian(A, None, lo, hi, 8, ne=8, fpm3=11, fpm4=loops)
        got = rci.rci_solve_hermitian(NumpyRciServer(A), lo, hi, 8, fpm_with(f2=8, f3=11, f4=loops))
        assert (got.info, got.M, got.loop) == (want.info, want.M, want.loop)
        assert got.loop == loops
        assert np.allclose(got.lambda_, want.lam, atol=1e-8)
        assert np.allclose(got.res, want.res, rtol=1e-3, atol=1e-10)


@pytest.mark.parametrize("generalized", [False, True])
def test_grci_matches_oracle(generalized):
    """feast_grci! with exact jobs == the oracle's variant-C loop.  For B != I the kernel's residual
    ||A q - lambda q|| (feast_kernel.jl:899-906, no B) cannot converge: all fpm[4] loops run, info 0."""
    n = 20
    rng = np.random.default_rng(3)
    T = np.diag(np.linspace(-3, 3, n) + 1j * rng.uniform(-1, 1, n)) + 0.1 * np.triu(rng.standard_normal((n, n)), 1)
    S = rng.standard_normal((n, n)) + n * np.eye(n)
    A = S @ T @ np.linalg.inv(S)
    B = np.diag(1.0 + rng.random(n)) if generalized else None
    ev = np.linalg.eigvals(A if B is None else np.linalg.solve(B, A))
    c = 0.3 + 0.1j
    dist = np.sort(np.abs(ev - c))
    r = 0.5 * (dist[6] + dist[7])                       # 7 eigenvalues inside, M0 = 10
    maxloop = 8 if generalized else 20
    want = fo.feast_general(A, B, c, r, 10, ne=16, fpm3=10, fpm4=maxloop)
    got = rci.rci_solve_general(NumpyRciServer(A, B), c, r, 10, fpm_with(f8=16, f3=10, f4=maxloop))
    assert (got.info, got.M) == (want.info, want.M)
    assert abs(got.loop - want.loop) <= 1              # stop test sits at rounding level for the last loop
    inside = ev[np.abs(ev - c) <= r]
    assert got.info == 0 and got.M == len(inside) == 7
    key = lambda x: (round(x.real, 6), round(x.imag, 6))
    assert np.allclose(sorted(got.lambda_, key=key), sorted(inside, key=key), atol=1e-8)
    assert np.allclose(sorted(got.lambda_, key=key), sorted(want.lam, key=key), atol=1e-10)
    if generalized:
        assert got.loop == maxloop and got.epsout > 1e-3
    else:
        assert got.epsout <= 1e-10 and want.epsout <= 1e-10     # converged: residuals are rounding noise


def test_grci_job_sequence():
    n = 10
    A = np.diag(np.arange(1.0, n + 1)) + 0.01 * np.triu(np.ones((n, n)), 1)
    srv = NumpyRciServer(A)
    r = rci.rci_solve_general(srv, 2.4 + 0j, 1.0, 4, fpm_with(f8=8, f3=9, f4=10))
    assert r.info == 0 and r.M == 2
    per_loop = [10, 11] * 8 + [40, 30, 30]
    assert srv.jobs == per_loop * (r.loop + 1)


# ---- the reference's matrix-free RCI fixtures (test/test_matrix_free.jl:55-185): the solver callback
# ---- receives ``work`` unmultiplied, residual ||A q - lambda q||
def test_matfree_small_symmetric_fixture():
    A = np.array([[2.0, -1, 0], [-1, 2, -1], [0, -1, 2]])
    r = rci.rci_solve_symmetric(NumpyRciServer(A), 0.5, 1.5, 3, fpm_with(f3=12, f4=20), matrix_free=True)
    assert r.info == 0 and r.M >= 1
    ev = np.linalg.eigvalsh(A)
    for lam in r.lambda_:
        assert np.min(np.abs(ev - lam)) < 1e-10
    assert np.all(r.res < 1e-8)


def test_matfree_tridiagonal_fixture():
    n = 100
    r = rci.rci_solve_symmetric(NumpyRciServer(tridiag(n)), 0.8, 1.2, 8, fpm_with(f3=8, f4=20), matrix_free=True)
    assert r.info == 0 and r.M > 0
    assert np.all((r.lambda_ >= 0.8) & (r.lambda_ <= 1.2))
    ev = 2 - 2 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))
    inside = ev[(ev >= 0.8) & (ev <= 1.2)]
    assert r.M == len(inside) and np.allclose(r.lambda_, inside, atol=1e-7)


def test_matfree_identity_operator_fixture():
    A = np.diag([1.0, 2.0, 3.0, 4.0, 5.0])
    r = rci.rci_solve_symmetric(NumpyRciServer(A), 1.5, 4.5, 5, fk.feastinit(), matrix_free=True)
    assert r.info == 0 and r.M == 3
    assert np.allclose(np.sort(r.lambda_), [2.0, 3.0, 4.0], atol=1e-10)


def test_linear_solver_callback_contract():
    """linear_solver(Y, z, X): Y = (zB - A)^-1 X for all columns (feast_matfree.jl:149)."""
    A = tridiag(7)
    B = np.diag(np.linspace(1, 2, 7))
    solve = NumpyRciServer(A, B).linear_solver()
    X = np.random.default_rng(0).standard_normal((7, 3))
    Y = np.zeros((7, 3), complex)
    z = 0.7 + 0.3j
    solve(Y, z, X)
    assert np.allclose((z * B - A) @ Y, X, atol=1e-12)


def test_user_subspace_fpm5():
    A = tridiag(16)
    Q0 = np.random.default_rng(1).standard_normal((16, 5))
    Q0[:, 2] = 0.0                                    # zero column -> random fallback, unit norm
    want_ev = np.linalg.eigvalsh(A)
    r = rci.rci_solve_symmetric(NumpyRciServer(A), 0.0, 0.6, 5, fpm_with(f3=10, f4=30), Q0=Q0)
    inside = want_ev[want_ev <= 0.6]
    assert r.info == 0 and r.M == len(inside) and np.allclose(r.lambda_, inside, atol=1e-9)
