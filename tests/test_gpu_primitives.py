"""GPU parity of every C-ABI primitive against numpy/scipy (the oracle's arithmetic),
through the ctypes binding.  Tolerances are fp64 round-off scaled by problem size."""
import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

import feast_oracle as fo
from kat_util import cmat, cplx as kcplx, load_kats

pytestmark = pytest.mark.gpu
K = load_kats()


def rand_block(N, m, seed, cplx=True):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, m))
    if cplx:
        X = X + 1j * rng.standard_normal((N, m))
    return np.asfortranarray(X.astype(np.complex128))


def sparse_pair(N, seed, cplx=False, b_identity=False):
    rng = np.random.default_rng(seed)
    A = sp.random(N, N, density=min(1.0, 6.0 / N), random_state=seed, format="csr")
    A = A + A.T + sp.diags(np.arange(1, N + 1, dtype=float))
    if cplx:
        S = sp.random(N, N, density=min(1.0, 3.0 / N), random_state=seed + 1, format="csr")
        A = A + 1j * (S - S.T)
    A = sp.csr_matrix(A)
    if b_identity:
        return A, None
    B = sp.random(N, N, density=min(1.0, 3.0 / N), random_state=seed + 2, format="csr")
    B = B + B.T + sp.diags(4.0 + rng.random(N))
    return A, sp.csr_matrix(B)


@pytest.mark.parametrize("N,m", [(50, 3), (257, 16), (1000, 17), (777, 32), (513, 48), (2049, 64), (640, 100)])
@pytest.mark.parametrize("cplx,bid", [(False, False), (True, False), (False, True)])
def test_spmm_matches_scipy(engine, N, m, cplx, bid):
    A, B = sparse_pair(N, 11 + N + m, cplx, bid)
    engine.set_problem(A, B)
    X = rand_block(N, m, 5)
    dX = engine.upload(X)
    YA = engine.download(engine.matmul(0, dX, m))
    YB = engine.download(engine.matmul(1, dX, m))
    refA = A @ X
    refB = X if B is None else B @ X
    assert np.abs(YA - refA).max() <= 1e-12 * max(1.0, np.abs(refA).max())
    assert np.abs(YB - refB).max() <= 1e-12 * max(1.0, np.abs(refB).max())


@pytest.mark.parametrize("N,m,cplx", [(777, 48, False), (1030, 64, True), (4100, 20, True)])
def test_dense_matmul_nonsymmetric(engine, N, m, cplx):
    """General (non-Hermitian) dense operands: a transposed or conjugated MFMA operand map would pass the
    Hermitian cases below; N is no multiple of the 16/32/64-row tiles and 4100 rows take the 64-row path."""
    rng = np.random.default_rng(N)
    A = rng.standard_normal((N, N)) + (1j * rng.standard_normal((N, N)) if cplx else 0)
    B = rng.standard_normal((N, N)) + (1j * rng.standard_normal((N, N)) if cplx else 0)
    engine.set_problem(A, B)
    X = rand_block(N, m, 8)
    dX = engine.upload(X)
    YA = engine.download(engine.matmul(0, dX, m))
    YB = engine.download(engine.matmul(1, dX, m))
    refA, refB = A @ X, B @ X
    assert np.abs(YA - refA).max() <= 1e-12 * np.abs(refA).max()
    assert np.abs(YB - refB).max() <= 1e-12 * np.abs(refB).max()


@pytest.mark.parametrize("N,m", [(64, 4), (300, 16), (1000, 33), (515, 64)])
@pytest.mark.parametrize("cplx,bid", [(False, False), (True, False), (False, True)])
def test_dense_matmul(engine, N, m, cplx, bid):
    rng = np.random.default_rng(N + m)
    A = rng.standard_normal((N, N)); A = A + A.T
    B = None
    if cplx:
        S = rng.standard_normal((N, N)); A = A + 1j * (S - S.T)
    if not bid:
        B = rng.standard_normal((N, N)); B = B @ B.T / N + np.eye(N)
    engine.set_problem(A, B)
    X = rand_block(N, m, 6)
    dX = engine.upload(X)
    YA = engine.download(engine.matmul(0, dX, m))
    YB = engine.download(engine.matmul(1, dX, m))
    refA, refB = A @ X, (X if B is None else B @ X)
    assert np.abs(YA - refA).max() <= 1e-11 * np.abs(refA).max()
    assert np.abs(YB - refB).max() <= 1e-11 * np.abs(refB).max()


@pytest.mark.parametrize("N,m", [(40, 5), (300, 16), (1000, 32), (2000, 64), (900, 65), (1500, 150)])   # m > 64: panels
def test_project_matches_numpy(engine, N, m):
    A, B = sparse_pair(N, 3, cplx=True)
    engine.set_problem(A, B)
    Q = rand_block(N, m, 9)
    dQ = engine.upload(Q)
    Aq, Bq = engine.project(dQ, m, bilinear=False, hermitize=True)
    refA = fo.hermitian_part(Q.conj().T @ (A @ Q))
    refB = fo.hermitian_part(Q.conj().T @ (B @ Q))
    assert np.abs(Aq - refA).max() <= 1e-11 * np.abs(refA).max()
    assert np.abs(Bq - refB).max() <= 1e-11 * np.abs(refB).max()
    Ar, Br = engine.project(dQ, m, bilinear=False, hermitize=False)
    assert np.abs(Ar - Q.conj().T @ (A @ Q)).max() <= 1e-11 * np.abs(refA).max()
    At, _ = engine.project(dQ, m, bilinear=True, hermitize=False)
    assert np.abs(At - Q.T @ (A @ Q)).max() <= 1e-11 * np.abs(refA).max()


@pytest.mark.parametrize("N,m", [(257, 16), (1000, 33), (2049, 64)])
def test_real_panels_take_the_one_product_path(engine, N, m):
    """Panels without imaginary parts (what a real projection of a real pencil produces) make the MFMA Gram and Q V kernels
    skip three of the four real products, decided from the loaded data.  The results must be those of the complex path:
    checked against numpy, and against the same call on the panel with ONE entry given a tiny imaginary part (which sends
    every workgroup that loads it down the four-product path)."""
    A, B = sparse_pair(N, 21, cplx=False)
    engine.set_problem(A, B)
    rng = np.random.default_rng(N + m)
    Q = rng.standard_normal((N, m)) + 0j
    dQ = engine.upload(Q)
    Aq, Bq = engine.project(dQ, m, bilinear=False, hermitize=False)
    refA, refB = Q.T @ (A @ Q), Q.T @ (B @ Q)
    assert not Aq.imag.any() and not Bq.imag.any()
    assert np.abs(Aq - refA).max() <= 1e-12 * np.abs(refA).max()
    assert np.abs(Bq - refB).max() <= 1e-12 * np.abs(refB).max()
    Q2 = Q.copy(); Q2[N // 2, m // 2] += 1e-300j
    A2, B2 = engine.project(engine.upload(Q2), m, bilinear=False, hermitize=False)
    assert np.array_equal(A2.real, Aq.real) and np.array_equal(B2.real, Bq.real)         # same real parts, bit for bit
    V = np.asfortranarray(rng.standard_normal((m, m)) + 0j)
    lam = np.linspace(-1.0, 1.0, m)
    dX, res = engine.ritz_residual(dQ, m, V, lam, m, normalize=False, use_B=True)
    X = engine.download(dX)[:, :m]
    assert not X.imag.any()
    assert np.abs(X - Q @ V).max() <= 1e-12 * np.abs(Q @ V).max()
    V2 = V.copy(); V2[0, 0] += 1e-300j
    dX2, _ = engine.ritz_residual(dQ, m, V2, lam, m, normalize=False, use_B=True)
    assert np.array_equal(engine.download(dX2)[:, :m].real, X.real)


@pytest.mark.parametrize("N,m,true_rank", [(30, 4, 4), (200, 16, 9), (1000, 32, 32), (3000, 64, 40), (500, 10, 3),
                                           (800, 100, 100), (1200, 130, 70), (600, 200, 64)])   # m > 64: block Gram-Schmidt
def test_orthonormalize_rank_and_span(engine, N, m, true_rank):
    A, B = sparse_pair(N, 3)
    engine.set_problem(A, B)
    rng = np.random.default_rng(N)
    basis = rng.standard_normal((N, true_rank)) + 1j * rng.standard_normal((N, true_rank))
    mix = rng.standard_normal((true_rank, m)) + 1j * rng.standard_normal((true_rank, m))
    src = np.asfortranarray(basis @ mix)
    dQ = engine.upload(src)
    rank = engine.orthonormalize(dQ, m, np.sqrt(np.finfo(float).eps))
    Qo, rank_ref = fo.qr_compress(src, m)
    assert rank == rank_ref == true_rank
    Q = engine.download(dQ)[:, :rank]
    assert np.abs(Q.conj().T @ Q - np.eye(rank)).max() < 1e-12
    # same span as the pivoted-QR basis of the oracle and reproduces the source
    assert np.linalg.norm(src - Q @ (Q.conj().T @ src)) <= 1e-11 * np.linalg.norm(src)
    assert np.linalg.norm(Qo - Q @ (Q.conj().T @ Qo)) <= 1e-10


def test_orthonormalize_reference_kat(engine):
    # rank-compress KAT of the reference (test/test_allocation_helpers.jl:274-292): the 4x4 literal whose fourth column
    # is 1e-15 (below the rank threshold max(sqrt(eps), eps*max(N, M0)) * |R_11|) and whose second is twice the first
    k = K["qr_compress"]
    A, B = sparse_pair(4, 3)
    engine.set_problem(A, B)
    src = cmat(k["src"])
    assert src[0, 3] == 1e-15 and src[1, 3] == 1e-15j
    dQ = engine.upload(src)
    rank = engine.orthonormalize(dQ, k["ncols"], np.sqrt(np.finfo(float).eps))
    assert rank == k["expect_rank"] == fo.qr_compress(src, k["ncols"])[1] == 2
    Q = engine.download(dQ)[:, :rank]
    assert np.abs(Q.conj().T @ Q - np.eye(rank)).max() < 1e-12
    assert np.linalg.norm(src - Q @ (Q.conj().T @ src)) <= k["span_tol"]


@pytest.mark.parametrize("kind", ["well", "graded", "ill"])
def test_orthonormalize_one_and_two_cholesky_qr_passes(kind, monkeypatch):
    """Cholesky-QR runs its second pass only when the pivot ratio of the equilibrated Gram matrix is below 1e-2.  "well":
    Gaussian columns (ratio ~ 1, one pass); "graded": the same columns scaled over four decades (equilibration makes it the
    first case: still one pass; ten decades would be rank deficient by the reference's |R_ii| / |R_11| rule); "ill": columns 3e-3 apart from each other in angle (ratio ~ 1e-5: two passes).  In every
    case the basis is orthonormal to 1e-12, spans the input, and equals -- up to 1e-10 in the projector -- what the
    always-two-passes form (FH_CHOLQR_TWO_PASS=1) returns."""
    import feastkit_jl_amd as fk
    N, m = 3000, 40
    rng = np.random.default_rng(77)
    X = rng.standard_normal((N, m)) + 1j * rng.standard_normal((N, m))
    if kind == "graded":
        X = X * np.logspace(-2, 2, m)[None, :]
    elif kind == "ill":
        X = X[:, :1] + 3e-3 * X
    A, B = sparse_pair(N, 5)
    out = {}
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("FH_CHOLQR_TWO_PASS", "1")
        eng = fk.HipEngine(0)
        eng.set_problem(A, B)
        dQ = eng.upload(X)
        rank = eng.orthonormalize(dQ, m, np.sqrt(np.finfo(float).eps))
        assert rank == m
        Q = eng.download(dQ)[:, :rank]
        eng.close()
        assert np.abs(Q.conj().T @ Q - np.eye(m)).max() < 1e-12
        assert np.linalg.norm(X - Q @ (Q.conj().T @ X)) <= 1e-9 * np.linalg.norm(X)
        out[forced] = Q
    # same subspace: Q1 Q1^H Q2 = Q2
    assert np.linalg.norm(out[False] @ (out[False].conj().T @ out[True]) - out[True]) < 1e-10


def test_moment_kat_on_device(engine):
    """Moment KAT of the reference (test/test_allocation_helpers.jl:219-265) through the device's want_moments path:
    one contour node z = Zne[1], weight Wne[1], B = I and a dense A built so that (z - A)^-1 work = workc, the
    literal solution block.  zAq must equal Wne[1]*(work'*workc) and zSq must equal Zne[1]*zAq."""
    k = K["moment_accumulation"]
    work, workc = cmat(k["work"]), cmat(k["workc"])
    w, z = kcplx(k["Wne1"]), kcplx(k["Zne1"])
    P = workc @ np.linalg.pinv(workc)
    S = work @ np.linalg.pinv(workc) + 3.0 * (np.eye(3) - P)
    A = z * np.eye(3) - S
    engine.set_problem(A, None)
    engine.set_contour(np.array([z]), np.array([w]), 1.0)
    engine.set_real_projection(False)
    engine.set_node_range(0, 1)
    engine.set_solver("direct")
    dP, status, _, zAq, zSq = engine.contour_apply(engine.upload(work), 2, None, want_moments=True)
    assert int(status[0]) == 0
    assert np.allclose(engine.download(dP, 2), w * workc, atol=1e-13)
    assert np.allclose(zAq, cmat(k["expect_Aq"]), atol=1e-13) and np.allclose(zSq, cmat(k["expect_Bq"]), atol=1e-13)
    oa, os_ = fo.node_moments(work, workc, w, z)
    assert np.allclose(zAq, oa, atol=1e-13) and np.allclose(zSq, os_, atol=1e-13)


@pytest.mark.parametrize("N,r,M", [(60, 6, 3), (400, 16, 16), (1500, 30, 11), (2500, 64, 44), (700, 100, 80), (1000, 129, 129)])
@pytest.mark.parametrize("use_B", [True, False])
def test_ritz_residual(engine, N, r, M, use_B):
    A, B = sparse_pair(N, 21)
    engine.set_problem(A, B)
    Q = rand_block(N, r, 2)
    V = rand_block(r, r, 3)
    lam = np.linspace(0.5, 3.0, r) + 0j
    dQ = engine.upload(Q)
    dX, res = engine.ritz_residual(dQ, r, V, lam, M, normalize=True, use_B=use_B)
    X = Q @ V
    X[:, :M] /= np.linalg.norm(X[:, :M], axis=0)
    Xg = engine.download(dX)
    assert np.abs(Xg - X).max() <= 1e-11 * np.abs(X).max()
    ref = fo.feast_residual(A, B if use_B else None, lam.real, X, M)
    assert np.abs(res - ref).max() <= 1e-10 * ref.max()


@pytest.mark.parametrize("N,m", [(200, 7), (1000, 32), (3000, 64)])
def test_bicgstab_shifted_solve(engine, N, m):
    # well-conditioned shifted system: z far from the (real) spectrum
    A, B = sparse_pair(N, 5)
    engine.set_problem(A, B)
    engine.set_solver("bicgstab", rtol=1e-12, atol=0.0, maxit=4000)
    z = -3.0 + 2.0j
    X = rand_block(N, m, 8)
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    assert rc == 0
    Y = engine.download(dY)
    S = (z * B - A).tocsc()
    ref = sp.linalg.splu(S).solve(X)
    rel = np.linalg.norm(S @ Y - X, axis=0) / np.linalg.norm(X, axis=0)
    assert rel.max() < 1e-10
    assert np.abs(Y - ref).max() <= 1e-8 * np.abs(ref).max()


@pytest.mark.parametrize("N,m", [(33, 5), (100, 10), (500, 32), (1111, 64)])
@pytest.mark.parametrize("cplx,bid", [(False, True), (False, False), (True, False)])
@pytest.mark.parametrize("prec", [64, 32])
def test_dense_lu_shifted_solve(engine, N, m, cplx, bid, prec):
    """prec 32: complex64 factors (f32 MFMA) + fp64 iterative refinement -- same accuracy bar as fp64 LU."""
    rng = np.random.default_rng(N)
    A = rng.standard_normal((N, N)); A = A + A.T
    if cplx:
        S = rng.standard_normal((N, N)); A = A + 1j * (S - S.T)
    B = None
    if not bid:
        B = rng.standard_normal((N, N)); B = B @ B.T / N + np.eye(N)
    engine.set_problem(A, B)
    engine.set_solver("direct", factor_precision=prec)
    z = 0.3 + 0.7j
    X = rand_block(N, m, 4)
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    engine.set_solver("direct")
    assert rc == 0
    Y = engine.download(dY)
    Sm = z * (np.eye(N) if B is None else B) - A
    ref = sla.lu_solve(sla.lu_factor(Sm), X)
    assert np.abs(Y - ref).max() <= 1e-9 * np.abs(ref).max()
    assert (np.linalg.norm(Sm @ Y - X, axis=0) / np.linalg.norm(X, axis=0)).max() < 1e-11


@pytest.mark.parametrize("kb,legacy_solve", [(64, False), (256, False), (128, True)])
@pytest.mark.parametrize("prec", [64, 32])
def test_dense_lu_block_widths(kb, legacy_solve, prec, monkeypatch):
    """The outer block column of the two-level LU is chosen by size (128, 256 from N = 6144); the 256-wide path and
    the 32-column substitution kept for comparison are run here on a small non-symmetric matrix whose size is no
    multiple of any block width.  The environment is read when the handle is created."""
    import feastkit_jl_amd as fk
    monkeypatch.setenv("FH_LU_KB", str(kb))
    if legacy_solve:
        monkeypatch.setenv("FH_LU_SOLVE_32", "1")
    eng = fk.HipEngine(0)
    N, m = 839, 40
    rng = np.random.default_rng(kb)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    eng.set_problem(A, None)
    eng.set_solver("direct", factor_precision=prec)
    z = 0.2 - 0.4j
    X = rand_block(N, m, 9)
    dY, rc = eng.shifted_solve(z, eng.upload(X), m)
    assert rc == 0
    Y = eng.download(dY)
    Sm = z * np.eye(N) - A
    assert (np.linalg.norm(Sm @ Y - X, axis=0) / np.linalg.norm(X, axis=0)).max() < 1e-10
    ref = sla.lu_solve(sla.lu_factor(Sm), X)
    assert np.abs(Y - ref).max() <= 1e-8 * np.abs(ref).max()
    eng.close()


@pytest.mark.parametrize("prec", [64, 32])
def test_dense_lu_lookahead_is_bit_identical(prec, monkeypatch):
    """The look-ahead (next block column's panels beside the rest of the trailing update, side stream with a CU mask,
    a plain low-priority stream, or a chunked rest update) reorders launches, not arithmetic: the solution of a shifted
    system must not change in a single bit against the serial order.  N = 1500 gives 12 outer blocks of 128."""
    import feastkit_jl_amd as fk
    N, m = 1500, 24
    rng = np.random.default_rng(15)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    z = 0.1 + 0.3j
    X = rand_block(N, m, 11)
    out = {}
    for name, env in (("serial", {"FH_LU_LOOKAHEAD": "0"}), ("mask", {}), ("mask2", {"FH_LU_RESERVE": "2"}),
                      ("plain", {"FH_LU_RESERVE": "0", "FH_LU_CHUNKS": "1"}), ("chunks", {"FH_LU_RESERVE": "0"})):
        for k in ("FH_LU_LOOKAHEAD", "FH_LU_RESERVE", "FH_LU_CHUNKS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = fk.HipEngine(0)
        eng.set_problem(A, None)
        eng.set_solver("direct", factor_precision=prec)
        dY, rc = eng.shifted_solve(z, eng.upload(X), m)
        assert rc == 0
        out[name] = eng.download(dY).copy()
        eng.close()
    Sm = z * np.eye(N) - A
    assert (np.linalg.norm(Sm @ out["serial"] - X, axis=0) / np.linalg.norm(X, axis=0)).max() < 1e-10
    for name in ("mask", "mask2", "plain", "chunks"):
        assert np.array_equal(out[name], out["serial"]), name


def test_dense_lu_trsm_product_matches_substitution(monkeypatch):
    """U block row by the product with L11^-1 (default) against the in-place substitution (FH_LU_TRSM_SUBST=1): same
    factorisation up to rounding, both against LAPACK."""
    import feastkit_jl_amd as fk
    N, m = 700, 16
    rng = np.random.default_rng(16)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    z = -0.2 + 0.5j
    X = rand_block(N, m, 12)
    Sm = z * np.eye(N) - A
    ref = sla.lu_solve(sla.lu_factor(Sm), X)
    for subst in (False, True):
        if subst:
            monkeypatch.setenv("FH_LU_TRSM_SUBST", "1")
        eng = fk.HipEngine(0)
        eng.set_problem(A, None)
        eng.set_solver("direct")
        dY, rc = eng.shifted_solve(z, eng.upload(X), m)
        assert rc == 0
        Y = eng.download(dY)
        eng.close()
        assert np.abs(Y - ref).max() <= 1e-9 * np.abs(ref).max()


@pytest.mark.parametrize("prec", [64, 32])
def test_dense_lu_tall_panels(engine, prec):
    """N > 8192: the first panels hold more than 8 x 1024 rows and take the 16-rows-per-thread panel kernel
    (k_lu_panel_reg<16,1>) and the 256-wide outer block; checked through the residual of a shifted solve."""
    N, m = 8500, 4
    rng = np.random.default_rng(85)
    A = (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))) / np.sqrt(N)
    A = np.asfortranarray(A)
    engine.set_problem(A, None)
    engine.set_solver("direct", factor_precision=prec)
    z = 2.5 + 0.5j                                   # outside the unit-disc spectrum: well conditioned
    X = rand_block(N, m, 3)
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    engine.set_solver("direct")
    assert rc == 0
    Y = engine.download(dY)
    R = z * Y - A @ Y - X
    assert (np.linalg.norm(R, axis=0) / np.linalg.norm(X, axis=0)).max() < 1e-11


def test_contour_apply_matches_oracle_sum(engine):
    # Q_proj = sum_e 2 w_e (z_e B - A)^{-1} B Q and the variant-B moments, dense LU path
    N, m = 120, 12
    rng = np.random.default_rng(0)
    A = rng.standard_normal((N, N)); A = A + A.T
    B = rng.standard_normal((N, N)); B = B @ B.T / N + np.eye(N)
    Zne, Wne = fo.feast_contour(-1.0, 1.0, 8)
    engine.set_problem(A, B)
    engine.set_contour(Zne, Wne, 2.0)
    engine.set_real_projection(False)      # complex half-contour sum (the session engine is shared)
    engine.set_solver("direct")
    Q = rand_block(N, m, 1, cplx=False)
    dP, status, stats, zAq, zSq = engine.contour_apply(engine.upload(Q), m, None, want_moments=True)
    assert (status[:8] == 0).all() and stats["factorizations"] == 8
    ref = np.zeros((N, m), complex); rA = np.zeros((m, m), complex); rS = np.zeros((m, m), complex)
    for z, w in zip(Zne, Wne):
        Y = np.linalg.solve(z * B - A, B @ Q)
        ref += 2 * w * Y
        rA += 2 * w * (Q.conj().T @ Y)
        rS += 2 * w * z * (Q.conj().T @ Y)
    assert np.abs(engine.download(dP) - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(zAq - rA).max() <= 1e-10 * np.abs(rA).max()
    assert np.abs(zSq - rS).max() <= 1e-10 * np.abs(rS).max()
    # node range = block partition: partial sums add up
    engine.set_node_range(0, 3)
    p0 = engine.download(engine.contour_apply(engine.upload(Q), m)[0])
    engine.set_node_range(3, 5)
    p1 = engine.download(engine.contour_apply(engine.upload(Q), m)[0])
    assert np.abs(p0 + p1 - ref).max() <= 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("m", [80, 100, 129])
def test_contour_apply_moments_wider_than_one_panel(engine, m):
    """Variant-B moments for M0 > 64: block columns of zAq/zSq assembled panel by panel (complex Hermitian input, the
    complex half-contour sum of src/parallel/feast_mpi.jl:564-567)."""
    N = 300
    rng = np.random.default_rng(4)
    H = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    A = np.diag(np.linspace(-3, 3, N)) + 0.05 * (H + H.conj().T)
    Zne, Wne = fo.feast_contour(-1.0, 1.0, 8)
    engine.set_problem(A, None)
    engine.set_contour(Zne, Wne, 2.0)
    engine.set_real_projection(False)
    engine.set_node_range(0, 8)
    engine.set_solver("direct")
    Q = rand_block(N, m, 2)
    dP, status, stats, zAq, zSq = engine.contour_apply(engine.upload(Q), m, None, want_moments=True)
    assert (status[:8] == 0).all()
    ref = np.zeros((N, m), complex); rA = np.zeros((m, m), complex); rS = np.zeros((m, m), complex)
    for z, w in zip(Zne, Wne):
        Y = np.linalg.solve(z * np.eye(N) - A, Q)
        ref += 2 * w * Y
        a, s_ = fo.node_moments(Q, Y, 2 * w, z)
        rA += a; rS += s_
    assert np.abs(engine.download(dP, m) - ref).max() <= 1e-10 * np.abs(ref).max()
    assert zAq.shape == (m, m)
    assert np.abs(zAq - rA).max() <= 1e-10 * np.abs(rA).max()
    assert np.abs(zSq - rS).max() <= 1e-10 * np.abs(rS).max()


@pytest.mark.parametrize("N,m", [(200, 7), (1000, 32), (3000, 64)])
def test_cocg_shifted_solve(engine, N, m):
    # complex-symmetric shifted system from real-symmetric A, B
    A, B = sparse_pair(N, 5)
    engine.set_problem(A, B)
    engine.set_solver("cocg", rtol=1e-12, atol=0.0, maxit=4000)
    z = -3.0 + 2.0j
    X = rand_block(N, m, 8)
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    assert rc == 0
    Y = engine.download(dY)
    S = (z * B - A).tocsc()
    rel = np.linalg.norm(S @ Y - X, axis=0) / np.linalg.norm(X, axis=0)
    assert rel.max() < 1e-10
    # complex Hermitian input is rejected (S would not be complex symmetric)
    Ac, Bc = sparse_pair(N, 5, cplx=True)
    engine.set_problem(Ac, Bc)
    import feastkit_jl_amd as fk
    with pytest.raises(fk.FeastHipError):
        engine.shifted_solve(z, engine.upload(X), m)
    engine.set_solver("direct")


@pytest.mark.parametrize("solver", ["bicgstab", "cocg"])
@pytest.mark.parametrize("N,m", [(300, 16), (2000, 64)])
def test_mixed_precision_correction_solve(engine, solver, N, m):
    """prec 32: complex64 Krylov correction on the fp64 residual; reaches the requested relative
    reduction (1e-4 here), verified against the fp64 residual on the host."""
    A, B = sparse_pair(N, 5)
    engine.set_problem(A, B)
    engine.set_solver(solver, rtol=1e-4, atol=0.0, maxit=2000, factor_precision=32)
    z = -3.0 + 2.0j
    X = rand_block(N, m, 8)
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    assert rc == 0
    Y = engine.download(dY)
    S = (z * B - A).tocsc()
    rel = np.linalg.norm(S @ Y - X, axis=0) / np.linalg.norm(X, axis=0)
    assert rel.max() < 3e-4
    engine.set_solver("direct")


@pytest.mark.parametrize("N,m,restart", [(200, 7, 30), (1000, 32, 20), (1500, 64, 40)])
@pytest.mark.parametrize("dense", [False, True])
def test_gmres_shifted_solve(engine, N, m, restart, dense):
    """Restarted GMRES(m) with the reference's stop test ||r|| <= atol + rtol ||r0||."""
    A, B = sparse_pair(N, 5)
    if dense:
        A, B = A.toarray(), B.toarray()
    engine.set_problem(A, B)
    engine.set_solver("gmres", rtol=1e-10, atol=1e-10, maxit=600, restart=restart)
    z = -3.0 + 2.0j
    X = rand_block(N, m, 8)
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    assert rc == 0
    Y = engine.download(dY)
    S = z * B - A
    rel = np.linalg.norm(S @ Y - X, axis=0) / np.linalg.norm(X, axis=0)
    assert rel.max() < 1e-9
    # same answer as the oracle's per-column restarted GMRES
    mv = lambda x: S @ x
    xo, ok, _ = fo.gmres_restarted(mv, X[:, 0], 1e-10, 1e-10, 600, restart)
    assert ok and np.abs(xo - Y[:, 0]).max() <= 1e-7 * np.abs(xo).max()
    engine.set_solver("direct")


@pytest.mark.parametrize("solver", ["cocg", "bicgstab"])
def test_contour_apply_is_bitwise_reproducible(engine, solver):
    """Two-stage fixed-order reductions and the in-order node sum of the COCG sum mode: the same call
    twice gives the same bits (no float atomics)."""
    import feastkit_jl_amd as fk
    A, B, lam = fo.cfg3_problem(12, 10, 9)
    N = A.shape[0]
    engine.set_problem(A, B)
    fpm = fk.feastdefault(fk.feastinit()); fpm[2] = 8
    Z, W = fk.feast_contour(0.0, 0.5, fpm)
    engine.set_contour(Z, W, 2.0)
    engine.set_real_projection(True)
    engine.set_solver(solver, rtol=1e-3, atol=0.0, maxit=60)
    Q = engine.upload(fk.seeded_subspace(N, 24))
    ritz = np.linspace(0.05, 0.9, 24)
    outs = []
    for _ in range(3):
        dP, status, st = engine.contour_apply(Q, 24, ritz)
        outs.append(engine.download(dP, 24).copy())
    engine.set_real_projection(False)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0


@pytest.mark.parametrize("N,r,generalized,cplx", [(60, 5, False, False), (300, 17, True, False), (800, 32, True, True),
                                                 (1500, 64, True, False), (700, 63, False, True)])
def test_rayleigh_ritz_on_device_matches_host_path(engine, N, r, generalized, cplx):
    """feasthip_rayleigh_ritz_dev (Jacobi eigensolver in LDS) against project + LAPACK zhegv + ritz_residual."""
    A, B = sparse_pair(N, 31, cplx=cplx, b_identity=not generalized)
    engine.set_problem(A, B)
    Q = rand_block(N, r, 4, cplx=cplx)
    dQ = engine.upload(Q)
    rank = engine.orthonormalize(dQ, r, 1e-8)
    assert rank == r
    Sq, Aq = engine.project(dQ, r, bilinear=False, hermitize=True)
    lam_ref, V = sla.eigh(Sq, Aq)
    lo, hi = lam_ref[r // 4] - 1e-9, lam_ref[(3 * r) // 4] + 1e-9
    inside = [i for i in range(r) if lo <= lam_ref[i] <= hi]
    perm = inside + [i for i in range(r) if i not in set(inside)]
    dXh, res_h = engine.ritz_residual(dQ, r, np.asfortranarray(V[:, perm]), lam_ref[perm], len(inside), normalize=True, use_B=True)
    out = engine.rayleigh_ritz(dQ, r, lo, hi, use_B=True)
    assert out is not None
    dX, lam, M, res = out
    scale = np.abs(lam_ref).max()
    assert M == len(inside)
    assert np.abs(lam - lam_ref[perm]).max() <= 1e-13 * scale
    assert np.abs(res - res_h).max() <= 1e-9 * max(res_h.max(), 1e-300) + 1e-13
    # same Ritz vectors up to a unit phase per column (eigenvalues are simple here)
    Xd, Xh = engine.download(dX)[:, :M], engine.download(dXh)[:, :M]
    for j in range(M):
        ph = np.vdot(Xh[:, j], Xd[:, j])
        assert abs(abs(ph) - 1.0) <= 1e-9
        assert np.linalg.norm(Xd[:, j] - ph * Xh[:, j]) <= 1e-8
    if not cplx:
        assert np.abs(engine.download(dX).imag).max() <= 1e-14 * np.abs(engine.download(dX)).max()   # real input stays real


def test_rayleigh_ritz_reports_indefinite_b(engine):
    N = 50
    A = sp.csr_matrix(sp.diags([np.arange(1.0, N + 1)], [0]))
    B = sp.csr_matrix(sp.diags([np.where(np.arange(N) % 2 == 0, 1.0, -1.0)], [0]))    # indefinite "B"
    engine.set_problem(A, B)
    dQ = engine.upload(rand_block(N, 6, 1, cplx=False))
    assert engine.orthonormalize(dQ, 6, 1e-8) == 6
    assert engine.rayleigh_ritz(dQ, 6, 0.0, 10.0) is None


def test_rayleigh_ritz_on_device_degenerate_spectrum(engine):
    """Repeated reduced eigenvalues (the Jacobi sweeps must still converge and keep V^H A V = I): check
    eigenvalues, residuals and B-orthonormality of the Ritz vectors instead of the vectors themselves."""
    N, r = 400, 24
    d = np.repeat(np.arange(1.0, 9.0), 50)                      # eigenvalues 1..8, each 50-fold
    A = sp.csr_matrix(sp.diags([d], [0]))
    B = sp.csr_matrix(sp.diags([1.0 + 0.01 * (np.arange(N) % 7)], [0]))
    engine.set_problem(A, B)
    rng = np.random.default_rng(3)
    idx = np.concatenate([rng.choice(np.where(d == v)[0], 3, replace=False) for v in range(1, 9)])   # 3 per cluster
    Q = np.zeros((N, r), dtype=np.complex128)
    Q[idx, np.arange(r)] = 1.0
    Q = Q @ np.linalg.qr(rng.standard_normal((r, r)) + 1j * rng.standard_normal((r, r)))[0]       # mix within the subspace
    dQ = engine.upload(np.asfortranarray(Q))
    out = engine.rayleigh_ritz(dQ, r, 2.5, 6.5, use_B=True)
    assert out is not None
    dX, lam, M, res = out
    bdiag = B.diagonal()
    want = np.sort(d[idx] / bdiag[idx])
    assert M == int(np.sum((want >= 2.5) & (want <= 6.5)))
    assert np.allclose(np.sort(lam), want, atol=1e-12)
    assert res.max() <= 1e-12
    X = engine.download(dX)[:, :M]
    G = X.conj().T @ (B @ X)
    assert np.abs(G - np.diag(np.diag(G))).max() <= 1e-12      # B-orthogonal (columns were normalised in 2-norm)
