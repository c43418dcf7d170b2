"""GPU tests of the sparse DIRECT solver for general patterns: reverse Cuthill-McKee renumbering + blocked band LU on the
dense MFMA kernels (fh_dense.hip fh_wband_*, behind FEASTHIP_SOLVER_BANDED).  The reference's counterpart is the sparse
drivers' default `lu(z B - A)` (UMFPACK, src/sparse/feast_sparse.jl:339-342); the checker here is SuperLU (scipy splu)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import feastkit_jl_amd as fk
from feastkit_jl_amd import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_blocked(monkeypatch):
    """the blocked band plan whatever the band width (and no multifrontal plan: tests/test_gpu_multifrontal.py has those)"""
    monkeypatch.setenv("FH_WBAND", "1")
    monkeypatch.setenv("FH_MF", "0")


def random_pencil(n, band, density, seed, cplx, symmetric_pattern):
    """A sparse matrix with entries inside |i - j| <= band at the given fill, no diagonal dominance (the row interchanges
    and the kl rows of fill are exercised), and a B on a narrower pattern."""
    rng = np.random.default_rng(seed)
    nnz = int(density * n * (2 * band + 1))
    i = rng.integers(0, n, nnz)
    j = np.clip(i + rng.integers(-band, band + 1, nnz), 0, n - 1)
    v = rng.standard_normal(nnz) + (1j * rng.standard_normal(nnz) if cplx else 0.0)
    A = sp.coo_matrix((v, (i, j)), shape=(n, n)).tocsr()
    if symmetric_pattern:
        A = A + A.T
    A = (A + sp.diags(rng.standard_normal(n) * 0.5)).tocsr()
    B = sp.diags([0.2 * rng.standard_normal(n - 1), 1.0 + 0.3 * rng.random(n), 0.2 * rng.standard_normal(n - 1)], [-1, 0, 1]).tocsr()
    return A, B


def check_solve(engine, A, B, z, m, seed=5, tol=2e-10):
    n = A.shape[0]
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    assert rc == 0
    Y = engine.download(dY, m)
    S = (z * (B if B is not None else sp.identity(n)) - A).tocsc().astype(complex)
    ref = spla.splu(S).solve(X)
    res = np.linalg.norm(S @ Y - X) / np.linalg.norm(X)
    res_ref = np.linalg.norm(S @ ref - X) / np.linalg.norm(X)
    assert res <= max(tol, 50 * res_ref), (res, res_ref)
    assert np.abs(Y - ref).max() <= 1e-6 * np.abs(ref).max()
    return Y


@pytest.mark.parametrize("n,band,density,m,cplx,sym", [
    (130, 9, 0.5, 5, True, False),            # fewer rows than one 128-block + its band
    (500, 40, 0.3, 16, True, False),          # unsymmetric pattern: kl != ku after the renumbering
    (1000, 150, 0.05, 64, False, True),
    (2100, 300, 0.02, 33, True, True),        # N not a multiple of 32: short last panel
    (3000, 600, 0.01, 64, False, False),      # panel rows > 1024 + 128: the two-rows-per-thread panel kernel
])
def test_blocked_band_solve_matches_superlu(engine, force_blocked, n, band, density, m, cplx, sym):
    A, B = random_pencil(n, band, density, 11 + n, cplx, sym)
    engine.set_problem(A, B)
    engine.set_solver("banded")
    kl, ku, nbytes, blocked = engine.band_plan()
    assert blocked == 1 and kl <= 2 * band + 2 and ku <= 2 * band + 2
    check_solve(engine, A, B, 0.3 + 0.8j, m)
    # identity B and a second shift through the cached-slot logic
    engine.set_problem(A, None)
    engine.set_solver("banded")
    check_solve(engine, A, None, -0.2 + 0.05j, m)


def test_blocked_band_takes_scrambled_order(engine, monkeypatch):
    """A narrow band hidden by a random symmetric permutation: the plan must find it again (reverse Cuthill-McKee) and
    the result must come back in the caller's order."""
    monkeypatch.setenv("FH_MF", "0")
    n = 3000
    A0, B0 = random_pencil(n, 20, 0.4, 3, True, True)
    p = np.random.default_rng(1).permutation(n)
    P = sp.identity(n, format="csr")[p]
    A = (P @ A0 @ P.T).tocsr()
    B = (P @ B0 @ P.T).tocsr()
    engine.set_problem(A, B)
    engine.set_solver("banded")
    kl, ku, nbytes, blocked = engine.band_plan()
    assert blocked == 1 and kl + ku <= 400, (kl, ku)          # stored order: ~ n; RCM: a small multiple of the hidden band
    check_solve(engine, A, B, 0.5 + 0.5j, 24)


def test_blocked_band_contour_apply_and_cache(engine, monkeypatch):
    """3-D stencil in lexicographic order (band 2 x 30 x 20, beyond the narrow-band window): the sweep of a half contour
    against SuperLU, factors cached across sweeps."""
    monkeypatch.setenv("FH_MF", "0")
    A, B, _ = workloads.laplacian_3d_pencil(30, 20, 12)
    n = A.shape[0]
    engine.set_problem(A, B)
    kl, ku, nbytes, blocked = engine.band_plan()
    assert blocked == 1 and kl == ku and kl < 600
    fpm = fk.feastdefault(fk.feastinit()); fpm[2] = 8
    Z, W = fk.feast_contour(0.0, 0.25, fpm)
    engine.set_contour(Z, W, 2.0)
    engine.set_real_projection(False)
    engine.set_solver("banded")
    Q = fk.seeded_subspace(n, 40)
    dP, status, st = engine.contour_apply(engine.upload(Q), 40)
    assert st["factorizations"] == 8 and np.all(status[:8] == 0)
    BQ = B @ Q
    want = sum(2 * W[e] * spla.splu((Z[e] * B - A).tocsc().astype(complex)).solve(BQ.astype(complex)) for e in range(8))
    got = engine.download(dP, 40)
    assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    dP2, status, st2 = engine.contour_apply(engine.upload(Q), 40)
    assert st2["factorizations"] == 0
    assert np.array_equal(engine.download(dP2, 40), got)


def test_blocked_band_singular_shift_reports_lapack(engine, force_blocked):
    n = 300
    A = sp.diags([np.arange(1.0, n + 1)], [0]).tocsr()
    engine.set_problem(A, None)
    engine.set_solver("banded")
    dY, rc = engine.shifted_solve(7.0 + 0j, engine.upload(np.ones((n, 2))), 2)
    assert rc == 8


def test_sparse_direct_interior_interval_full_size():
    """cfg 3's pencil, an interval deep inside the spectrum: the Krylov sweeps cannot solve these shifted systems (DESIGN
    section 5), the direct solver -- the reference's default for sparse input -- converges in a few loops."""
    A, B, lam = workloads.laplacian_3d_pencil(50, 40, 25)
    mid = 2.0
    order = np.argsort(np.abs(lam - mid))
    r = 0.5 * (abs(lam[order[39]] - mid) + abs(lam[order[40]] - mid))
    inside = np.sort(lam[np.abs(lam - mid) < r])
    assert inside.size == 40
    fpm = fk.feastinit(); fpm[2] = 8
    res = fk.feast(A, B, (mid - r, mid + r), M0=64, fpm=fpm, solver="banded")
    assert res.info == 0 and res.M == 40
    assert np.abs(np.sort(res.lambda_) - inside).max() <= 1e-10
    X = res.q[:, :res.M]
    R = A @ X - (B @ X) * res.lambda_[:res.M]
    assert (np.linalg.norm(R, axis=0) / np.maximum(np.abs(res.lambda_[:res.M]), 1.0)).max() <= 1e-10


def test_default_call_falls_back_to_the_direct_solver(engine, monkeypatch):
    """`feast(A, B, interval)` with every keyword at its default, on an interval inside the spectrum: the Krylov sweeps the
    default maps to (beyond the band / dense windows) stop with info = 5; solver=:direct was what the caller asked for, so
    the call then runs the sparse direct solver and returns what the reference's default would have."""
    monkeypatch.setattr(fk.api, "_DIRECT_FLOPS", 0.0)                     # (this small pencil would be served directly at once: the fallback is the subject)
    A, B, lam = workloads.laplacian_3d_pencil(30, 24, 18)                 # N = 12 960: beyond the dense window
    mid = 2.0
    order = np.argsort(np.abs(lam - mid))
    r = 0.5 * (abs(lam[order[19]] - mid) + abs(lam[order[20]] - mid))
    inside = np.sort(lam[np.abs(lam - mid) < r])
    assert inside.size == 20
    assert fk.api._sparse_direct_solver(A, B, 8) == "krylov"
    fpm = fk.feastinit()                  # (fpm[4] = 20 loops: the time extrapolation ends the Krylov attempt after a few)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = fk.feast(A, B, (mid - r, mid + r), M0=32, fpm=fpm, engine=engine)
    sub = res.stats["solver_substitution"]
    assert sub["used"] == "cocg" and sub["fallback"].split()[0] in ("band", "multifrontal") and sub["krylov_info"] == 5 and sub["krylov_loops"] <= 8
    assert res.info == 0 and res.M == 20
    assert np.abs(np.sort(res.lambda_) - inside).max() <= 1e-10


def test_blocked_band_fuzz(engine, force_blocked):
    """Seeded random shapes through the blocked band LU: size, band, fill, right-hand sides, value type, pattern symmetry,
    B present or not -- every case against SuperLU."""
    rng = np.random.default_rng(20260515)
    for case in range(14):
        n = int(rng.integers(140, 2600))
        band = int(rng.integers(3, min(420, n // 3)))
        density = float(rng.uniform(0.02, 0.6)) * min(1.0, 40.0 / band)
        m = int(rng.integers(1, 65))
        cplx = bool(rng.integers(0, 2))
        sym = bool(rng.integers(0, 2))
        A, B = random_pencil(n, band, density, 1000 + case, cplx, sym)
        if rng.integers(0, 3) == 0:
            B = None
        engine.set_problem(A, B)
        engine.set_solver("banded")
        z = complex(rng.uniform(-0.5, 0.5), rng.uniform(0.05, 1.0))
        try:
            check_solve(engine, A, B, z, m, seed=case)
        except AssertionError as exc:
            raise AssertionError(f"case {case}: n={n} band={band} density={density:.3f} m={m} cplx={cplx} sym={sym} B={'I' if B is None else 'tri'}: {exc}")


def test_blocked_band_complex64_factors_refined(engine, force_blocked):
    """factor_precision = 32: complex64 band factors, every solve refined in fp64 against the CSR operator -- the result
    has fp64 accuracy (the dense solver's mixed-precision mode, on the sparse direct solver)."""
    A, B = random_pencil(1800, 120, 0.1, 77, True, False)
    engine.set_problem(A, B)
    engine.set_solver("banded", rtol=1e-13, factor_precision=32)
    kl, ku, nbytes64, blocked = engine.band_plan()
    Y = check_solve(engine, A, B, 0.4 + 0.6j, 48, tol=5e-12)
    assert engine.last_stats["max_rel_residual"] <= 1e-12
    # without refinement (rtol >= 1) the same factors give single precision only
    engine.set_solver("banded", rtol=1.0, factor_precision=32)
    n, m = A.shape[0], 48
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))
    dY, rc = engine.shifted_solve(0.4 + 0.6j, engine.upload(X), m)
    S = (0.4 + 0.6j) * B - A
    res32 = np.linalg.norm(S @ engine.download(dY, m) - X) / np.linalg.norm(X)
    assert rc == 0 and 1e-9 < res32 < 1e-1, res32             # cond x eps32 on this pencil, no more


def test_sparse_direct_mixed_precision_full_size():
    """cfg 3 with complex64 band factors + fp64 refinement: same eigenpairs, half the factor memory."""
    A, B, lam = workloads.laplacian_3d_pencil(50, 40, 25)
    inside = np.sort(lam[(lam >= 0.0) & (lam <= 0.1775)])
    fpm = fk.feastinit(); fpm[2] = 16
    res = fk.feast(A, B, (0.0, 0.1775), M0=64, fpm=fpm, solver="banded", inner_precision=32)
    assert res.info == 0 and res.M == 44
    assert np.abs(np.sort(res.lambda_) - inside).max() <= 1e-11
    X = res.q[:, :res.M]
    R = A @ X - (B @ X) * res.lambda_[:res.M]
    assert (np.linalg.norm(R, axis=0) / np.maximum(np.abs(res.lambda_[:res.M]), 1.0)).max() <= 1e-11


def test_sparse_general_direct_beyond_the_dense_window():
    """The sparse GENERAL driver with every keyword at its default on a non-symmetric 3-D convection-diffusion operator
    (N = 17 920, 7-point pattern: beyond the dense window), the eigenvalues in a disc at the lower end of the spectrum.  The
    default `solver=:direct` is served by the sparse direct solver (16 full-contour factorisations of the reordered band);
    checker: ARPACK shift-invert on SuperLU factors."""
    nx, ny, nz = 32, 28, 20

    def d2(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")

    def d1(n):
        return sp.diags([-np.ones(n - 1), np.ones(n - 1)], [-1, 1], format="csr")          # central difference: skew-symmetric
    Ix, Iy, Iz = sp.identity(nx), sp.identity(ny), sp.identity(nz)
    lap = sp.kron(Iz, sp.kron(Iy, d2(nx))) + sp.kron(Iz, sp.kron(d2(ny), Ix)) + sp.kron(d2(nz), sp.kron(Iy, Ix))
    conv = 0.3 * sp.kron(Iz, sp.kron(Iy, d1(nx))) + 0.2 * sp.kron(Iz, sp.kron(d1(ny), Ix))
    A = sp.csr_matrix(lap + conv)
    assert abs(A - A.T).max() > 0.1 and fk.api._sparse_direct_solver(A, None, 16) == "krylov"
    ref = spla.eigs(A.tocsc().astype(complex), k=32, sigma=0.0, which="LM", return_eigenvectors=False, tol=1e-12)
    ref = ref[np.argsort(np.abs(ref))]
    radius = 0.5 * (abs(ref[20]) + abs(ref[21]))                   # the widest gap nearby: 21 eigenvalues inside
    inside = ref[np.abs(ref) < radius]
    assert inside.size == 21
    fpm = fk.feastinit(); fpm[8] = 16
    res = fk.feast_general(A, None, 0.0, radius, M0=inside.size, fpm=fpm)
    assert res.stats["solver_substitution"]["used"].split()[0] in ("band", "multifrontal") and res.stats["factorizations"] == 16
    assert res.info == 0 and res.M == inside.size
    key = lambda z: (round(z.real, 8), round(z.imag, 8))
    assert np.allclose(sorted(res.lambda_, key=key), sorted(inside, key=key), atol=1e-9)
    R = A @ res.q - res.q * res.lambda_
    assert (np.linalg.norm(R, axis=0) / np.linalg.norm(res.q, axis=0)).max() <= 1e-10
