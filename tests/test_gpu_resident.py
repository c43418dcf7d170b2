"""GPU tests of the refinement loop with resident panels (feasthip_contour_apply_resident / rr_reduce_resident /
rr_ritz_resident / resident_export): every stage against the per-primitive entry points on the same data and against
numpy, the implicit-basis fast path and the rank-revealing general path, column blocks, and the drivers end to end with
the resident loop against the per-primitive loop (same M, info, loop count, eigenvalues)."""
import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

import feast_oracle as fo
import feastkit_jl_amd as fk
from test_gpu_primitives import rand_block, sparse_pair

pytestmark = pytest.mark.gpu
SQRT_EPS = float(np.sqrt(np.finfo(float).eps))


def contour8(engine, lo=2.0, hi=9.0, ne=8, scale=2.0):
    fpm = fk.feastinit(); fpm[2] = ne
    fk.feastdefault(fpm)
    Z, W = fk.feast_contour(lo, hi, fpm)
    engine.set_contour(Z, W, scale)
    return Z, W


def oracle_sweep(A, B, Q, Z, W, scale, real_part):
    N = A.shape[0]
    Bm = sp.identity(N, format="csc") if B is None else sp.csc_matrix(B)
    rhs = Bm @ Q
    P = np.zeros_like(Q)
    for z, w in zip(Z, W):
        P += scale * w * sp.linalg.splu(sp.csc_matrix(z * Bm - A)).solve(rhs)
    return P.real.astype(np.complex128) if real_part else P


@pytest.mark.parametrize("N,m,bid,real_part", [(700, 24, False, True), (900, 64, False, False), (800, 40, True, True), (333, 7, False, False)])
def test_resident_stages_match_primitives(engine, N, m, bid, real_part):
    import scipy.sparse.linalg  # noqa: F401  (sp.linalg)
    A, B = sparse_pair(N, 5, cplx=False, b_identity=bid)
    engine.set_problem(A, B)
    Z, W = contour8(engine)
    engine.set_real_projection(real_part)
    engine.set_node_range(0, len(Z))
    engine.set_solver("bicgstab", rtol=1e-13, atol=0.0, maxit=4000)
    Q = rand_block(N, m, 3, cplx=not real_part)
    dQ = engine.upload(Q)
    # --- sweep: resident Q_proj == the per-primitive Q_proj == the oracle's sum ---
    dP_ref, st_ref, _ = engine.contour_apply(dQ, m)
    status, _ = engine.contour_apply_resident(dQ, m)
    assert (status[:len(Z)] == 0).all() and (st_ref[:len(Z)] == 0).all()
    P_res = engine.download(engine.export_resident(m, which=1))
    P_ref = engine.download(dP_ref)
    want = oracle_sweep(A, B, Q, Z, W, 2.0, real_part)
    assert np.abs(P_res - P_ref).max() <= 1e-13 * np.abs(P_ref).max()
    assert np.abs(P_res - want).max() <= 1e-9 * np.abs(want).max()
    # --- reduction: rank, reduced pencil of an orthonormal basis of range(Q_proj) ---
    rank, Sq, Aq = engine.rr_reduce_resident(m, SQRT_EPS)
    _, rank_ref = fo.qr_compress(P_ref, m)
    assert rank == rank_ref            # (full rank in three of the cases; 22 of 40 behind the 8-node filter of the B = I case)
    Bd = np.eye(N) if B is None else B.toarray()
    lam, V = sla.eigh(Sq, Aq)
    if bid and rank < m:
        assert np.array_equal(Aq, np.eye(rank))      # orthonormal basis of the general path, B = I: exactly I
    # the per-primitive path on the same panel: the same rank and the same Ritz values
    dPo = dP_ref.clone()
    assert engine.orthonormalize(dPo, m, SQRT_EPS) == rank
    S2, A2 = engine.project(dPo, rank)
    assert np.abs(sla.eigh(S2, A2, eigvals_only=True) - lam).max() <= 1e-9 * max(1.0, np.abs(lam).max())
    Qref = engine.download(dPo)[:, :rank]
    lam_ref = sla.eigh(Qref.conj().T @ (A @ Qref), Qref.conj().T @ (Bd @ Qref), eigvals_only=True)
    assert np.abs(lam - lam_ref).max() <= 1e-9 * max(1.0, np.abs(lam_ref).max())
    # --- Ritz step: X = Q_o V, normalised, residuals; then X is the next sweep's subspace ---
    m_in, m = m, rank                  # from here on the block has `rank` columns
    M = max(1, m // 2)
    res = engine.rr_ritz_resident(m, V, lam, M)
    X = engine.download(engine.export_resident(m))
    # Ritz vectors: in range(Q_proj), first M of unit length, and they diagonalise the pencil
    assert np.allclose(np.linalg.norm(X[:, :M], axis=0), 1.0, atol=1e-12)
    want_res = fo.feast_residual(A, B, lam, X, M)
    assert np.abs(res - want_res).max() <= 1e-9 * max(want_res.max(), 1e-300) + 1e-13
    # (a basis cut at the rank threshold sqrt(eps) is defined up to directions of that size: two runs agree to ~1e-7 there)
    assert np.linalg.norm(X - Qref @ (Qref.conj().T @ X)) <= (1e-9 if rank == m_in else 1e-6) * np.linalg.norm(X)
    G = X.conj().T @ (A @ X) - (X.conj().T @ (Bd @ X)) * lam[None, :]
    assert np.abs(G).max() <= (1e-8 if rank == m_in else 1e-6) * max(1.0, np.abs(lam).max()) * np.abs(X.conj().T @ (Bd @ X)).max()
    # --- the next sweep from the resident Ritz block (with and without the warm start) == the sweep of the exported block ---
    for lam_guess in (None, lam.copy()):
        dX = engine.export_resident(m).clone()
        dP2, _, _ = engine.contour_apply(dX, m, lam_guess)
        engine.contour_apply_resident(None, m, lam_guess)
        P2 = engine.download(engine.export_resident(m, which=1))
        assert np.abs(P2 - engine.download(dP2)).max() <= 1e-10 * np.abs(engine.download(dP2)).max()
        # leave the Ritz block in place for the second round
        rank2, S3, A3 = engine.rr_reduce_resident(m, SQRT_EPS)
        l3, V3 = sla.eigh(S3, A3)
        engine.rr_ritz_resident(rank2, V3, l3, min(M, rank2))
        m, lam = rank2, l3                 # (the rank may drop again: the block then has rank2 columns)


def test_resident_general_path_rank_deficient_and_ill_conditioned(engine):
    """Q_proj of rank 9 in 20 columns, and a full-rank but ill-conditioned one (columns 3e-3 apart in angle: the implicit
    basis is refused, the two-pass Cholesky-QR of the general path takes over): rank as the oracle's pivoted QR, reduced
    pencil and Ritz vectors of the orthonormal basis."""
    N, m = 900, 20
    A, B = sparse_pair(N, 9)
    engine.set_problem(A, B)
    Bd = B.toarray()
    rng = np.random.default_rng(4)
    for kind in ("deficient", "ill"):
        if kind == "deficient":
            src = (rng.standard_normal((N, 9)) + 1j * rng.standard_normal((N, 9))) @ (rng.standard_normal((9, m)) + 0j)
        else:
            G = rng.standard_normal((N, m)) + 1j * rng.standard_normal((N, m))
            src = G[:, :1] + 3e-3 * G
        src = np.asfortranarray(src)
        engine.import_resident(engine.upload(src), m, which=1)          # the block as Q_proj
        rank, Sq, Aq = engine.rr_reduce_resident(m, SQRT_EPS)
        _, rank_ref = fo.qr_compress(src, m)
        assert rank == rank_ref == (9 if kind == "deficient" else m)
        lam, V = sla.eigh(Sq, Aq)
        Qo = sla.orth(src, rcond=1e-10) if kind == "deficient" else np.linalg.qr(src)[0]
        assert Qo.shape[1] == rank
        lam_ref = sla.eigh(Qo.conj().T @ (A @ Qo), Qo.conj().T @ (Bd @ Qo), eigvals_only=True)
        assert np.abs(lam - lam_ref).max() <= 1e-8 * np.abs(lam_ref).max()
        res = engine.rr_ritz_resident(rank, V, lam, rank)
        X = engine.download(engine.export_resident(rank))
        assert np.abs(res - fo.feast_residual(A, B, lam, X, rank)).max() <= 1e-9 * max(1.0, res.max())
        assert np.linalg.norm(X - Qo @ (Qo.conj().T @ X)) <= 1e-8 * np.linalg.norm(X)


def test_resident_column_block_one_rank(engine):
    """A column block on a single rank: the resident Q_proj holds the swept columns and zeros elsewhere, as the
    per-primitive sweep does."""
    N, m = 800, 48
    A, B = sparse_pair(N, 5)
    engine.set_problem(A, B)
    Z, W = contour8(engine)
    engine.set_real_projection(True)
    engine.set_node_range(0, len(Z))
    engine.set_solver("bicgstab", rtol=1e-13, atol=0.0, maxit=4000)
    dQ = engine.upload(rand_block(N, m, 6, cplx=False))
    try:
        engine.set_column_block(16, 16)
        dP, _, _ = engine.contour_apply(dQ, m)
        engine.contour_apply_resident(dQ, m)
    finally:
        engine.set_column_block(0, -1)
    P = engine.download(engine.export_resident(m, which=1))
    Pr = engine.download(dP)
    assert np.abs(P - Pr).max() <= 1e-13 * np.abs(Pr).max() and np.abs(P[:, :16]).max() == 0 and np.abs(P[:, 32:]).max() == 0
    assert np.abs(P[:, 16:32]).max() > 0


@pytest.mark.parametrize("case", ["sparse_krylov_inexact", "dense_direct", "sparse_exact_bicgstab", "hermitian_complex"])
def test_drivers_resident_loop_equals_per_primitive_loop(engine, case):
    """feast_hip_hermitian with the resident loop (default) and with resident_panels=False: same M, info, loop count, eigenvalues
    to 1e-11, residuals below the tolerance in both."""
    if case == "dense_direct":
        N = 768                                   # cfg 2's shape (tests/test_gpu_feast.py::test_cfg2_reduced_dense_lu)
        A = fo.householder_conjugated_diag(0.01 * np.arange(N)); B = None
        kw = dict(solver="direct")
        lo, hi, M0 = 0.995, 1.195, 32
    elif case == "hermitian_complex":
        N = 300
        rng = np.random.default_rng(2)
        G = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        A = np.diag(np.linspace(0, 30, N)) + 0.05 * (G + G.conj().T); B = None
        kw = dict(solver="direct", real_projection=False)
        ev = np.linalg.eigvalsh(A)
        lo, hi, M0 = 0.5 * (ev[40] + ev[41]), 0.5 * (ev[52] + ev[53]), 24
    else:
        A, B, lam_exact = fo.cfg3_problem(14, 12, 9)
        lo, hi, M0 = 0.0, 0.55, 40
        kw = (dict(solver="cocg", warm_start=True, inner_rtol=3e-2, solver_maxiter=100) if case == "sparse_krylov_inexact"
              else dict(solver="bicgstab", solver_tol=1e-13, solver_maxiter=4000, warm_start=False))
    out = []
    for res_on in (True, False):
        fpm = fk.feastinit(); fpm[2] = 8; fpm[4] = 60
        out.append(fk.feast_hip_hermitian(engine, A, B, lo, hi, M0, fpm, resident_panels=res_on, **kw))
    a, b = out
    assert a.info == b.info == 0 and a.M == b.M > 0 and abs(a.loop - b.loop) <= 1
    assert np.abs(np.sort(a.lambda_) - np.sort(b.lambda_)).max() <= 1e-11 * max(1.0, np.abs(b.lambda_).max())
    assert a.epsout <= 1e-12 and b.epsout <= 1e-12
    Bd = None if B is None else B
    for r in (a, b):
        Ax = A @ r.q
        Bx = r.q if Bd is None else Bd @ r.q
        hres = np.linalg.norm(Ax - Bx * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
        assert hres.max() <= 1e-10


@pytest.mark.parametrize("N,m,kind", [(500, 12, "real"), (600, 24, "real"), (450, 40, "real_bid"), (520, 64, "real"), (640, 64, "real_bid")])
def test_lazy_start_matches_materialised_start(engine, monkeypatch, N, m, kind):
    """The COCG sweep that never writes its start residual / direction (first product reads the shared source panel times
    per-node column factors: both gather kernels, every panel width) against the same sweep with the start materialised (FH_NO_LAZY_START) and against sparse LU."""
    import scipy.sparse.linalg  # noqa: F401
    A, B = sparse_pair(N, 21 + m, cplx=False, b_identity=(kind == "real_bid"))
    engine.set_problem(A, B)
    Z, W = contour8(engine)
    real_part = True
    engine.set_real_projection(real_part)
    engine.set_node_range(0, len(Z))
    engine.set_solver("cocg", rtol=1e-12, atol=0.0, maxit=3000)
    Q = rand_block(N, m, 8, cplx=not real_part)
    dQ = engine.upload(Q)
    lam = np.linspace(2.2, 8.8, m) if real_part else None          # warm start x0 = q / (z - lambda) per column
    out = {}
    for mode in ("lazy", "materialised"):
        if mode == "materialised":
            monkeypatch.setenv("FH_NO_LAZY_START", "1")
        dP, st, stats = engine.contour_apply(dQ, m, ritz_lambda=lam)
        assert (st[:len(Z)] == 0).all()
        status, _ = engine.contour_apply_resident(dQ, m, ritz_lambda=lam)
        assert (status[:len(Z)] == 0).all()
        out[mode] = (engine.download(dP), engine.download(engine.export_resident(m, which=1)), stats["krylov_iterations"])
    monkeypatch.delenv("FH_NO_LAZY_START")
    want = oracle_sweep(A, B, Q, Z, W, 2.0, real_part)
    scale = np.abs(want).max()
    for mode in out:
        assert np.abs(out[mode][0] - want).max() <= 1e-9 * scale
        assert np.abs(out[mode][1] - want).max() <= 1e-9 * scale
    assert np.abs(out["lazy"][0] - out["materialised"][0]).max() <= 1e-9 * scale      # both stop at rtol 1e-12 of their own recurrences
    assert abs(out["lazy"][2] - out["materialised"][2]) <= max(2, out["lazy"][2] // 10)     # same iterates to rounding: columns stop within a few steps of each other
