"""GPU tests of the multifrontal sparse direct solver (csrc/fh_mf.hpp symbolic phase, fh_dense.hip fh_mf_* numeric phase,
behind FEASTHIP_SOLVER_BANDED like the band LU it replaces when its work is smaller).  The reference's counterpart is the
sparse drivers' default `lu(z B - A)` (UMFPACK, src/sparse/feast_sparse.jl:334-342); the checker is SuperLU (scipy splu)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import feastkit_jl_amd as fk
from feastkit_jl_amd import workloads
from test_gpu_wband import check_solve, random_pencil

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_mf(monkeypatch):
    monkeypatch.setenv("FH_MF", "1")


def grid_pencil(nx, ny, nz, seed, cplx=False, unsym=False):
    """3-D 7-point stencil with random coefficients (no diagonal dominance), optionally unsymmetric values / complex."""
    rng = np.random.default_rng(seed)
    n = nx * ny * nz
    idx = np.arange(n).reshape(nx, ny, nz)
    rows, cols = [], []
    for a, b in ((idx[1:], idx[:-1]), (idx[:, 1:], idx[:, :-1]), (idx[:, :, 1:], idx[:, :, :-1])):
        rows += [a.ravel(), b.ravel()]
        cols += [b.ravel(), a.ravel()]
    r = np.concatenate(rows + [np.arange(n)])
    c = np.concatenate(cols + [np.arange(n)])
    v = rng.standard_normal(len(r)) + (1j * rng.standard_normal(len(r)) if cplx else 0.0)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    if not unsym:
        A = (A + A.T) * 0.5
    B = sp.diags([0.1 * rng.standard_normal(n - 1), 1.0 + 0.3 * rng.random(n), 0.1 * rng.standard_normal(n - 1)], [-1, 0, 1]).tocsr()
    return A.tocsr(), B


@pytest.mark.parametrize("shape,m,cplx,unsym,leaf", [
    ((6, 5, 4), 5, False, False, None),        # a handful of fronts
    ((12, 10, 8), 16, True, False, None),
    ((20, 16, 9), 64, False, True, None),      # unsymmetric values on a symmetric pattern
    ((25, 20, 12), 33, True, True, "24"),      # small leaves: deep tree, many groups
    ((30, 30, 1), 40, False, False, "200"),    # 2-D, large leaves
    ((34, 26, 12), 64, False, False, None),    # pivot blocks beyond 256: the 128-block inverses
])
def test_multifrontal_solve_matches_superlu(engine, force_mf, monkeypatch, shape, m, cplx, unsym, leaf):
    if leaf:
        monkeypatch.setenv("FH_MF_LEAF", leaf)
    A, B = grid_pencil(*shape, seed=sum(shape) + m, cplx=cplx, unsym=unsym)
    engine.set_problem(A, B)
    engine.set_solver("banded")
    kl, ku, nbytes, blocked = engine.band_plan()
    assert blocked == 2
    check_solve(engine, A, B, 0.3 + 0.8j, m)
    # a second shift through the cached-slot logic, then identity B
    check_solve(engine, A, B, -0.4 + 0.3j, m, seed=9)
    engine.set_problem(A, None)
    engine.set_solver("banded")
    assert engine.band_plan()[3] == 2
    check_solve(engine, A, None, -0.2 + 0.05j, m)


@pytest.mark.parametrize("n,band,density,m,cplx,sym", [
    (500, 40, 0.3, 16, True, False),           # unsymmetric PATTERN: the plan works on pattern U pattern^T
    (2100, 300, 0.02, 33, True, True),         # random band graph: poor separators, big fronts
    (3000, 20, 0.4, 24, False, True),
])
def test_multifrontal_random_patterns_match_superlu(engine, force_mf, n, band, density, m, cplx, sym):
    A, B = random_pencil(n, band, density, 23 + n, cplx, sym)
    if n == 3000:                              # hidden by a random symmetric permutation
        p = np.random.default_rng(1).permutation(n)
        P = sp.identity(n, format="csr")[p]
        A, B = (P @ A @ P.T).tocsr(), (P @ B @ P.T).tocsr()
    engine.set_problem(A, B)
    engine.set_solver("banded")
    assert engine.band_plan()[3] == 2
    check_solve(engine, A, B, 0.3 + 0.8j, m)


def test_multifrontal_contour_apply_cache_and_reproducibility(engine, force_mf, monkeypatch):
    A, B, _ = workloads.laplacian_3d_pencil(30, 20, 13)       # (not the band test's 30 x 20 x 12: the engine keeps the plan of a matrix it already holds)
    n = A.shape[0]
    engine.set_problem(A, B)
    assert engine.band_plan()[3] == 2
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
    # factor again from scratch: no atomics anywhere, so bit for bit the same
    engine.free_factors()
    dP3, status, st3 = engine.contour_apply(engine.upload(Q), 40)
    assert st3["factorizations"] == 8
    assert np.array_equal(engine.download(dP3, 40), got)
    # node batches (long contours: fronts x nodes is a grid dimension): 3 + 3 + 2 nodes per call, the same bits again
    monkeypatch.setenv("FH_MF_NODES_PER_CALL", "3")
    engine.free_factors()
    dP4, status, st4 = engine.contour_apply(engine.upload(Q), 40)
    assert st4["factorizations"] == 8 and np.all(status[:8] == 0)
    assert np.array_equal(engine.download(dP4, 40), got)


def test_multifrontal_complex64_factors_refined(engine, force_mf):
    """factor_precision = 32 on the multifrontal plan: complex64 fronts (half the store), every solve refined in fp64 against the
    CSR operator -- fp64 accuracy; without refinement the same factors give single precision only."""
    A, B = grid_pencil(24, 18, 10, seed=77, cplx=True, unsym=True)
    engine.set_problem(A, B)
    engine.set_solver("banded", rtol=1e-13, factor_precision=32)
    kl, ku, nbytes64, blocked = engine.band_plan()
    assert blocked == 2
    check_solve(engine, A, B, 0.4 + 0.6j, 48, tol=5e-12)
    assert engine.last_stats["max_rel_residual"] <= 1e-12
    engine.set_solver("banded", rtol=1.0, factor_precision=32)
    n, m = A.shape[0], 48
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))
    dY, rc = engine.shifted_solve(0.4 + 0.6j, engine.upload(X), m)
    S = (0.4 + 0.6j) * B - A
    res32 = np.linalg.norm(S @ engine.download(dY, m) - X) / np.linalg.norm(X)
    assert rc == 0 and 1e-9 < res32 < 1e-2, res32
    # back to complex128 factors on the same handle: the slots are re-made
    engine.set_solver("banded", rtol=1e-12, factor_precision=64)
    check_solve(engine, A, B, 0.4 + 0.6j, 48, seed=6)


def test_multifrontal_singular_shift_reports_lapack(engine, force_mf):
    n = 300
    A = sp.diags([np.arange(1.0, n + 1)], [0]).tocsr()
    engine.set_problem(A, None)
    engine.set_solver("banded")
    assert engine.band_plan()[3] == 2
    dY, rc = engine.shifted_solve(7.0 + 0j, engine.upload(np.ones((n, 2))), 2)
    assert rc == 8


def test_cfg3_takes_the_multifrontal_plan(engine):
    """BASELINE cfg 3 at full size: the library's own choice is the multifrontal plan (a tenth of the band's work, under
    0.8 GB of factors per node against 2.77 GB), and the reference's default call through it finds the 44 eigenpairs."""
    A, B, lam = workloads.laplacian_3d_pencil(50, 40, 25)
    engine.set_problem(A, B)
    kl, ku, nbytes, blocked = engine.band_plan()
    assert blocked == 2 and nbytes < 0.8e9
    assert engine.direct_plan_flops() < 0.12 * 8.0 * A.shape[0] * kl * (kl + ku)
    fpm = fk.feastinit(); fpm[2] = 16
    r = fk.feast(A, B, (0.0, 0.1775), M0=64, fpm=fpm, solver="banded", engine=engine)
    inside = lam[(lam > 0.0) & (lam < 0.1775)]
    assert r.info == 0 and r.M == len(inside)
    assert np.abs(np.sort(r.lambda_[:r.M]) - np.sort(inside)).max() < 1e-10
    X = r.q[:, :r.M]
    res = np.linalg.norm(A @ X - (B @ X) * r.lambda_[:r.M], axis=0) / np.linalg.norm(X, axis=0)
    assert res.max() < 1e-10


def test_multifrontal_fuzz(engine, force_mf):
    """Seeded random cases through the multifrontal solver -- grids with random holes (irregular separators), random band
    graphs, disconnected unions; size, leaf size, right-hand sides, value type, B present or not -- every case against SuperLU."""
    import os
    rng = np.random.default_rng(20261005)
    for case in range(14):
        kind = int(rng.integers(0, 3))
        cplx = bool(rng.integers(0, 2))
        m = int(rng.integers(1, 65))
        os.environ["FH_MF_LEAF"] = str(int(rng.choice([8, 16, 32, 64, 128])))
        try:
            if kind == 0:                                            # grid with holes
                shape = (int(rng.integers(4, 22)), int(rng.integers(4, 18)), int(rng.integers(1, 9)))
                A, B = grid_pencil(*shape, seed=500 + case, cplx=cplx, unsym=bool(rng.integers(0, 2)))
                n = A.shape[0]
                keep = np.sort(rng.choice(n, size=max(40, int(n * rng.uniform(0.6, 1.0))), replace=False))
                A, B = A[keep][:, keep].tocsr(), B[keep][:, keep].tocsr()
                desc = f"grid {shape} keep {len(keep)}"
            elif kind == 1:                                          # random band graph
                n = int(rng.integers(140, 2600))
                band = int(rng.integers(3, min(200, n // 3)))
                A, B = random_pencil(n, band, float(rng.uniform(0.02, 0.5)) * min(1.0, 30.0 / band), 900 + case, cplx, bool(rng.integers(0, 2)))
                desc = f"band n={n} band={band}"
            else:                                                    # two unconnected blocks
                A1, B1 = grid_pencil(int(rng.integers(3, 12)), int(rng.integers(3, 12)), int(rng.integers(1, 6)), seed=700 + case, cplx=cplx)
                A2, B2 = random_pencil(int(rng.integers(50, 600)), 12, 0.3, 800 + case, cplx, True)
                A, B = sp.block_diag([A1, A2], format="csr"), sp.block_diag([B1, B2], format="csr")
                desc = f"blocks {A1.shape[0]} + {A2.shape[0]}"
            if rng.integers(0, 3) == 0:
                B = None
            engine.set_problem(A, B)
            engine.set_solver("banded")
            assert engine.band_plan()[3] == 2
            z = complex(rng.uniform(-0.5, 0.5), rng.uniform(0.05, 1.0))
            try:
                check_solve(engine, A, B, z, m, seed=case)
            except AssertionError as exc:
                raise AssertionError(f"case {case}: {desc} m={m} cplx={cplx} leaf={os.environ['FH_MF_LEAF']} B={'I' if B is None else 'tri'}: {exc}")
        finally:
            del os.environ["FH_MF_LEAF"]


def _residual_check(engine, A, B, z, m=32, tol=1e-11):
    n = A.shape[0]
    engine.set_problem(A, B)
    engine.set_solver("banded")
    assert engine.band_plan()[3] == 2
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    assert rc == 0
    Y = engine.download(dY, m)
    S = z * (B if B is not None else sp.identity(n)) - A
    res = np.linalg.norm(S @ Y - X) / np.linalg.norm(X)
    assert res <= tol, res


def test_multifrontal_large_patterns_by_residual(engine, force_mf):
    """Sizes the host checker cannot follow (no SuperLU here: the residual of the solve in fp64 on the host): a 2-D grid of
    360 000 unknowns, and an unstructured 2-D mesh stand-in (6 nearest neighbours of 120 000 random points) in a scrambled
    numbering -- separators found on a graph without any grid structure."""
    A, B, _ = workloads.laplacian_3d_pencil(600, 600, 1)
    _residual_check(engine, A, B, 0.3 + 0.2j)
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(3)
    n = 120000
    pts = rng.random((n, 2))
    _, idx = cKDTree(pts).query(pts, k=7)
    rows = np.repeat(np.arange(n), 6)
    W = sp.coo_matrix((-np.ones(len(rows)), (rows, idx[:, 1:].ravel())), shape=(n, n)).tocsr()
    W = ((W + W.T) * 0.5).tocsr()
    A = (W + sp.diags(-np.asarray(W.sum(axis=1)).ravel() + 0.1)).tocsr()
    p = rng.permutation(n)
    A = A[p][:, p].tocsr()
    _residual_check(engine, A, None, 0.05 + 0.1j)
