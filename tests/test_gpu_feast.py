"""End-to-end parity of the :hip backend (through the C ABI, on the GPU) against the CPU
oracle and the reference's own known-answer fixtures.  Parity is asserted on converged
quantities only (eigenvalues, residuals, M, info): the reference's initial subspace comes
from Julia's MersenneTwister and is not reproducible."""
import numpy as np
import pytest
import scipy.sparse as sp

import feast_oracle as fo
import feastkit_jl_amd as fk
from kat_util import cmat, cplx, load_kats, sparse_tridiag, tridiag

pytestmark = pytest.mark.gpu
K = load_kats()


def fpm_with(**kw):
    fpm = fk.feastinit()
    for k, v in kw.items():
        fpm[int(k[1:])] = v
    return fpm


def test_tridiag3_real_symmetric(engine):
    k = K["tridiag3_real_sym"]
    r = fk.feast(tridiag(3), np.eye(3), tuple(k["interval"]), M0=3, engine=engine)
    assert r.info == 0 and r.M == 3 and not np.iscomplexobj(r.q)
    assert np.allclose(np.sort(r.lambda_), sorted(k["expect_lambda"]), atol=k["atol"])


@pytest.mark.parametrize("name", ["hermitian3_dense", "hermitian3_sparse"])
def test_hermitian3(engine, name):
    k = K[name]
    A = cmat(k["A"])
    if name.endswith("sparse"):
        r = fk.feast(sp.csr_matrix(A), None, tuple(k["interval"]), M0=3, engine=engine, solver="bicgstab", solver_maxiter=200)
    else:
        r = fk.feast(A, None, tuple(k["interval"]), M0=3, engine=engine)
    assert r.info == 0 and r.M == 3
    assert np.allclose(np.sort(r.lambda_), k["expect_lambda"], atol=k["atol"])
    # eigenvectors: residual against the full matrix on the host
    for j in range(3):
        x = r.q[:, j]
        assert np.linalg.norm(A @ x - r.lambda_[j] * x) <= 1e-9 * np.linalg.norm(x)


def test_general2_dense(engine):
    k = K["general2"]
    A, B = cmat(k["A"]), cmat(k["B"])
    r = fk.feast_general(A, None, cplx(k["center"]), k["radius"], M0=2, engine=engine)
    assert r.info == 0 and r.M == 2 and np.allclose(np.sort(r.lambda_.real), k["expect_standard"], atol=k["atol"])
    r = fk.feast_general(A, B, cplx(k["center"]), k["radius"], M0=2, engine=engine)
    assert r.info == 0 and r.M == 2 and np.allclose(np.sort(r.lambda_.real), k["expect_generalized"], atol=k["atol"])
    r = fk.feast_general(np.array([[1.0, 2.0], [0.0, 3.0]]), None, 2.0, 2.5, M0=2, engine=engine)   # real promotion
    assert r.info == 0 and r.M == 2 and np.allclose(np.sort(r.lambda_.real), [1.0, 3.0], atol=1e-9)


def test_diag80_oversized_subspace_rank_compression(engine):
    k = K["diag80_oversized"]
    A = np.diag(np.arange(1.0, 81))
    fpm = fpm_with(f2=k["fpm2"], f3=k["fpm3"], f4=k["fpm4"])
    for real_projection in (False, True):
        r = fk.feast(A, None, tuple(k["interval"]), M0=k["M0"], fpm=fpm.copy(), engine=engine, real_projection=real_projection)
        assert r.info == 0 and r.M == 2
        assert np.allclose(np.sort(r.lambda_), k["expect_lambda"], atol=k["atol"]) and r.res.max() < k["max_res"]
    ref = fo.feast_hermitian(A, None, *k["interval"], k["M0"], ne=8, fpm3=7, fpm4=4)
    assert ref.M == 2


def test_tridiag10_all_partitions_agree(engine):
    k = K["tridiag10_backends"]
    A = sparse_tridiag(10)
    fpm = fpm_with(f2=8, f4=20)
    r = fk.feast(A.toarray(), None, tuple(k["interval"]), M0=10, fpm=fpm.copy(), engine=engine)
    assert r.info == 0 and np.allclose(np.sort(r.lambda_), k["expect_lambda"], atol=k["atol"])
    rs = fk.feast(A, None, tuple(k["interval"]), M0=10, fpm=fpm.copy(), engine=engine, solver="bicgstab", solver_maxiter=500)
    assert rs.info == 0 and np.allclose(np.sort(rs.lambda_), k["expect_lambda"], atol=k["atol"])


def test_hermitian_generalized_diag6(engine):
    k = K["hermitian_generalized_diag6"]
    A = sp.diags(np.array(k["A_diag"], dtype=complex)).tocsr()
    B = sp.diags(np.array(k["B_diag"], dtype=complex)).tocsr()
    r = fk.feast(A, B, tuple(k["interval"]), M0=6, engine=engine, solver="bicgstab")
    assert r.info == 0 and r.M == len(k["expect_lambda"])
    assert np.allclose(np.sort(r.lambda_), k["expect_lambda"], atol=k["atol"])
    rd = fk.feast(A.toarray(), B.toarray(), tuple(k["interval"]), M0=6, engine=engine)
    assert rd.info == 0 and np.allclose(np.sort(rd.lambda_), k["expect_lambda"], atol=k["atol"])


def test_iterative_equals_direct_tridiag12(engine):
    k = K["gmres_equiv_tridiag12"]
    A = sparse_tridiag(12)
    d = fk.feast(A.toarray(), None, tuple(k["interval"]), M0=12, engine=engine)
    g = fk.feast(A, None, tuple(k["interval"]), M0=12, engine=engine, solver="bicgstab", solver_tol=k["solver_tol"],
                 solver_maxiter=k["maxiter"])
    assert d.info == 0 and g.info == 0 and d.M == g.M == len(k["expect_lambda"])
    assert np.allclose(np.sort(g.lambda_), np.sort(d.lambda_), atol=k["atol"])


def test_gmres_keyword_equals_direct_tridiag12(engine):
    """solver=:gmres with the reference's keywords (solver_tol, maxiter, restart), runtests.jl:552-580."""
    k = K["gmres_equiv_tridiag12"]
    A = sparse_tridiag(12)
    d = fk.feast(A.toarray(), None, tuple(k["interval"]), M0=12, engine=engine)
    g = fk.feast(A, None, tuple(k["interval"]), M0=12, engine=engine, solver="gmres", solver_tol=k["solver_tol"],
                 solver_maxiter=k["maxiter"], solver_restart=k["restart"])
    assert d.info == 0 and g.info == 0 and d.M == g.M == len(k["expect_lambda"])
    assert np.allclose(np.sort(g.lambda_), np.sort(d.lambda_), atol=k["atol"])
    gd = fk.feast(A.toarray(), None, tuple(k["interval"]), M0=12, engine=engine, solver="gmres", solver_tol=1e-8,
                  solver_maxiter=400, solver_restart=20, fpm=fpm_with(f3=8))
    assert gd.info == 0 and np.allclose(np.sort(gd.lambda_), np.sort(d.lambda_), atol=1e-8)


def test_mpi_complex_fixtures(engine):
    k = K["mpi_complex_hermitian_diag4"]
    A = sp.diags(np.array(k["diag"], dtype=complex)).tocsr()
    r = fk.feast(A, sp.identity(4, dtype=complex, format="csr"), tuple(k["interval"]), M0=4, fpm=fpm_with(f2=8, f4=12),
                 engine=engine, solver="bicgstab")
    assert r.info == 0 and np.allclose(np.sort(r.lambda_), k["expect_lambda"], atol=k["atol"])
    g = K["mpi_complex_general_diag4"]
    Ag = np.diag([cplx(v) for v in g["diag"]])
    r = fk.feast_general(Ag, np.eye(4, dtype=complex), cplx(g["center"]), g["radius"], M0=4,
                         fpm=fpm_with(f3=g["fpm3"], f4=g["fpm4"], f8=g["fpm8"]), engine=engine)
    want = sorted((cplx(v) for v in g["expect_lambda"]), key=lambda x: (x.real, x.imag))
    assert r.info == 0 and r.M == 3
    assert np.allclose(sorted(r.lambda_, key=lambda x: (round(x.real, 10), round(x.imag, 10))), want, atol=g["atol"])


def test_variant_a_loop_for_loop_vs_oracle(engine):
    """Same Q0, complex half-contour sum, direct solves: the GPU walks the reference's variant A
    with identical M / loop / info and eigenvalues to 1e-10."""
    N, M0 = 150, 24
    A = fo.householder_conjugated_diag(0.05 * np.arange(N))
    rng = np.random.default_rng(3)
    u = rng.random(N)
    H = np.eye(N) - 2 * np.outer(u, u) / (u @ u)
    B = H @ np.diag(1 + 0.5 * rng.random(N)) @ H
    B = 0.5 * (B + B.T)
    Q0 = fo.seeded_subspace(N, M0)
    ref = fo.feast_hermitian(A, B, 1.0, 1.3, M0, ne=8, fpm4=60, Q0=Q0)
    got = fk.feast_hip_hermitian(engine, A, B, 1.0, 1.3, M0, fpm_with(f2=8, f4=60), solver="direct", real_projection=False, Q0=Q0)
    assert (got.info, got.M) == (ref.info, ref.M) == (0, 7) and abs(got.loop - ref.loop) <= 1 and ref.loop > 10
    assert np.allclose(np.sort(got.lambda_), np.sort(ref.lam), atol=1e-10)
    assert got.epsout <= 1e-12 and ref.epsout <= 1e-12


def test_cfg1_readme_quickstart(engine):
    """BASELINE cfg 1: n=100 tridiagonal Laplacian, (0.5,1.5).  M0=10 is undersized (19
    eigenvalues inside): like the reference the solver must NOT report success.  M0=30 returns all 19."""
    A = tridiag(100)
    exact = np.array([2 - 2 * np.cos(k * np.pi / 101) for k in range(1, 101)])
    inside = exact[(exact >= 0.5) & (exact <= 1.5)]
    assert len(inside) == 19
    r10 = fk.feast(A, None, (0.5, 1.5), M0=10, engine=engine)
    o10 = fo.feast_hermitian(A, None, 0.5, 1.5, 10)
    assert r10.info != 0 and o10.info != 0
    r30 = fk.feast(A, None, (0.5, 1.5), M0=30, engine=engine)
    o30 = fo.feast_hermitian(A, None, 0.5, 1.5, 30, fpm4=60)
    assert r30.info == 0 and r30.M == 19 == o30.M
    assert np.allclose(np.sort(r30.lambda_), inside, atol=1e-10) and np.allclose(np.sort(o30.lam), inside, atol=1e-10)
    assert r30.epsout <= 1e-12


def test_cfg2_reduced_dense_lu(engine):
    """cfg 2 shape at N=768: A = H2 H1 diag(0.01 i) H1 H2, 20 eigenvalues 1.00..1.19 inside."""
    N = 768
    A = fo.householder_conjugated_diag(0.01 * np.arange(N))
    r = fk.feast(A, None, (0.995, 1.195), M0=32, fpm=fpm_with(f2=8), engine=engine)
    want = 0.01 * np.arange(100, 120)
    assert r.info == 0 and r.M == 20 and np.allclose(np.sort(r.lambda_), want, atol=1e-10)
    assert r.epsout <= 1e-12 and r.stats["factorizations"] == 8          # LU cached per node across loops
    true_res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.linalg.norm(r.q, axis=0)
    assert true_res.max() <= 1e-10


def test_cfg3_reduced_reference_mode_and_fast_mode(engine):
    """cfg 3 shape on a 16x12x10 grid: (i) reference semantics (zero initial guess, tol 1e-12
    per loop), (ii) warm-started inexact solves.  Both must match the closed form to 1e-10."""
    A, B, lam = fo.cfg3_problem(16, 12, 10)
    Emin, Emax = 0.0, 0.42
    inside = lam[(lam >= Emin) & (lam <= Emax)]
    M0 = len(inside) + 12
    strict = fk.feast(A, B, (Emin, Emax), M0=M0, fpm=fpm_with(f2=8), engine=engine, solver="bicgstab",
                      solver_tol=1e-12, solver_maxiter=3000, warm_start=False)
    fast = fk.feast(A, B, (Emin, Emax), M0=M0, fpm=fpm_with(f2=8, f4=40), engine=engine, solver="bicgstab",
                    warm_start=True, inner_rtol=1e-2, solver_maxiter=100)
    # the reference's variant A as written stops with M=0/info=5 at loop 0 here (half-contour
    # filter on a random start); the oracle with the real-part projection is the CPU answer
    assert fo.feast_hermitian(A, B, Emin, Emax, M0, ne=8).info == 5
    ref = fo.feast_hermitian(A, B, Emin, Emax, M0, ne=8, real_projection=True)
    assert ref.info == 0
    mixed = fk.feast(A, B, (Emin, Emax), M0=M0, fpm=fpm_with(f2=8, f4=40), engine=engine, solver="bicgstab",
                     warm_start=True, inner_rtol=1e-2, solver_maxiter=100, inner_precision=32)
    # guard columns (Ritz value outside the interval) keep their warm start after loop 2: same answer
    fpm = fpm_with(f2=8, f4=40)
    frozen = fk.feast_hip_hermitian(engine, A, B, Emin, Emax, M0, fpm, solver="cocg", warm_start=True, inner_rtol=1e-2,
                                    solver_maxiter=100, real_projection=True, freeze_guards_after=2)
    # reduced eigenproblem on the device (Jacobi) instead of host LAPACK: same answer, same loop count
    devrr = fk.feast_hip_hermitian(engine, A, B, Emin, Emax, M0, fpm_with(f2=8, f4=40), solver="cocg", warm_start=True,
                                   inner_rtol=1e-2, solver_maxiter=100, real_projection=True, reduced_solver="device")
    hostrr = fk.feast_hip_hermitian(engine, A, B, Emin, Emax, M0, fpm_with(f2=8, f4=40), solver="cocg", warm_start=True,
                                    inner_rtol=1e-2, solver_maxiter=100, real_projection=True)
    assert abs(devrr.loop - hostrr.loop) <= 1
    for r in (strict, fast, mixed, frozen, devrr):
        assert r.info == 0 and r.M == len(inside) == ref.M
        assert np.allclose(np.sort(r.lambda_), inside, atol=1e-10)
        assert np.allclose(np.sort(r.lambda_), np.sort(ref.lam), atol=1e-10)
        assert r.epsout <= 1e-12
        res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
        assert res.max() <= 1e-10        # residual recomputed in fp64 on the host


def _cfg3_check(r, A, B, inside):
    assert r.info == 0 and r.M == 44 and r.epsout <= 1e-12
    assert np.abs(np.sort(r.lambda_) - inside).max() <= 1e-10
    res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
    assert res.max() <= 1e-12 * 10       # residual recomputed in fp64 on the host


def test_cfg3_reference_default_call_full_size(engine):
    """The reference's call for the headline workload with NO keyword beyond the problem definition:
    feast(A, B, (Emin, Emax); M0=64, fpm[2]=16).  The reference dispatches sparse input to its direct solver
    (src/core/feast_backend_utils.jl:166-198 -> src/sparse/feast_sparse.jl:334-342); the :hip backend maps that default
    to its converging Krylov configuration (COCG, Ritz warm start, inexact solves, real projection) and must return all
    44 eigenpairs of cfg 3 at full size."""
    A, B, lam = fo.cfg3_problem(50, 40, 25)
    inside = lam[(lam >= 0.0) & (lam <= 0.1775)]
    r = fk.feast(A, B, (0.0, 0.1775), M0=64, fpm=fpm_with(f2=16), engine=engine)
    assert not np.iscomplexobj(r.q)
    _cfg3_check(r, A, B, inside)
    assert r.loop <= 12
    # fpm[18] was left unset: the driver chose the ellipse ratio itself (contour policy), taller than the circle
    pol = r.stats["contour_policy"]["fpm18_per_loop"]
    assert r.stats["solver_substitution"]["contour_policy"] == "auto" and min(pol) > 100 and len(pol) == r.loop + 1
    # a caller who SETS fpm[18] (even to the reference's default value) keeps it: the reference's contour, the reference's answer
    rc = fk.feast(A, B, (0.0, 0.1775), M0=64, fpm=fpm_with(f2=16, f18=100), engine=engine)
    _cfg3_check(rc, A, B, inside)
    assert "contour_policy" not in rc.stats
    assert rc.stats["krylov_iterations"] > 2 * r.stats["krylov_iterations"]      # what the policy is for


@pytest.mark.parametrize("switch", ["1", "0"])
def test_default_call_interior_interval_full_size(engine, switch, monkeypatch):
    """The default call on an INTERIOR interval of cfg 3's pencil (40 eigenvalues around 0.52: guards on both sides,
    indefinite shifted systems).  With the Krylov path alone (FEASTKIT_DIRECT_SWITCH=0) the contour policy still delivers
    every eigenpair (measured: 1.8 s, 14 loops, against 9.2 s on the circle).  By default the call extrapolates the time the
    Krylov loops still need after each of them and, here, hands over to the sparse direct solver after four (1.3 s).  Deep
    inside the spectrum, where 40 eigenvalues span 1e-2 at lambda = 2, no unpreconditioned Krylov sweep converges on any
    contour: tests/test_gpu_wband.py."""
    monkeypatch.setenv("FEASTKIT_DIRECT_SWITCH", switch)
    A, B, lam = fo.cfg3_problem(50, 40, 25)
    i0 = int(np.searchsorted(lam, 0.5))
    lo, hi = 0.5 * (lam[i0 - 1] + lam[i0]), 0.5 * (lam[i0 + 39] + lam[i0 + 40])
    r = fk.feast(A, B, (lo, hi), M0=64, fpm=fpm_with(f2=16, f4=40), engine=engine)
    assert r.info == 0 and r.M == 40 and np.abs(np.sort(r.lambda_) - lam[i0:i0 + 40]).max() <= 1e-10
    res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
    assert res.max() <= 1e-11
    sub = r.stats["solver_substitution"]
    if switch == "0":
        assert "fallback" not in sub and min(r.stats["contour_policy"]["fpm18_per_loop"]) > 100
    else:
        assert sub.get("fallback", "").split()[0] in ("band", "multifrontal") and sub["krylov_loops"] <= 6 and r.loop <= 3


_PENCILS = {"diag_mass_3d": ((50, 40, 25), "diag_mass"), "stiff_mass_3d": ((50, 40, 25), "stiff_mass"), "diag_mass_2d": ((400, 125), "diag_mass")}


@pytest.mark.parametrize("name", sorted(_PENCILS))
def test_default_call_noncommuting_pencils_full_size(engine, name):
    """The default call beyond cfg 3: three N = 50 000 pencils whose A and B do NOT commute (variable-coefficient
    diffusion operators in 3-D and 2-D with a random lumped mass matrix or a second, independently weighted operator as
    B; no closed-form spectrum).  CPU answer: scipy's shift-invert Lanczos (ARPACK + SuperLU) on the same matrices.
    feast(A, B, (Emin, Emax); M0=64, fpm[2]=16) must return every eigenvalue of the interval with residuals
    recomputed on the host <= 1e-11, with the contour policy engaged (3-D) or through the sparse direct solver (2-D: cheap factorisations)."""
    import scipy.sparse.linalg as spla
    dims, kind = _PENCILS[name]
    A, B = fk.workloads.variable_coefficient_pencil(dims, kind)
    assert abs(A @ B - B @ A).max() > 1.0                       # far from commuting
    assert fk.api._sparse_direct_solver(A, B, 16) == "krylov"   # (a 250 x 200 grid would be a band narrow enough for the banded LU)
    w = np.sort(spla.eigsh(A, k=48, M=B, sigma=0.0, which="LM", return_eigenvectors=False, tol=1e-12))
    Emax = 0.5 * (w[43] + w[44])
    r = fk.feast(A, B, (0.0, Emax), M0=64, fpm=fpm_with(f2=16, f4=40), engine=engine)
    assert r.info == 0 and r.M == 44 and r.epsout <= 1e-12, (r.info, r.M, r.epsout, r.loop)
    assert np.abs(np.sort(r.lambda_) - w[:44]).max() <= 1e-9 * w[43]
    res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0) / np.linalg.norm(r.q, axis=0)
    assert res.max() <= 1e-11
    used = (r.stats.get("solver_substitution") or {}).get("used", "")
    if used.split()[0] in ("multifrontal", "band"):
        # the 2-D pencil: all 16 factorisations together are under the direct solver's flop threshold (api._DIRECT_FLOPS), the
        # default `solver=:direct` is then served directly -- exact solves, two or three loops
        assert name == "diag_mass_2d" and r.loop <= 3 and r.stats["krylov_iterations"] == 0
    else:
        pol = r.stats["contour_policy"]["fpm18_per_loop"]
        assert min(pol) > 100 and r.loop <= 16


@pytest.mark.parametrize("aspect,cap", [(4000, 50), (100, 100)])
def test_cfg3_bench_settings_full_size(engine, aspect, cap):
    """The path bench.py times, at full size: COCG fp64 in sum mode (no per-node solution panels), Ritz warm start,
    inner rtol 3e-2, balanced node lists, column groups by rule, on the bench's contour (Gauss nodes on the ellipse
    fpm[18] = 4000) and on the reference's default circle."""
    A, B, lam = fo.cfg3_problem(50, 40, 25)
    inside = lam[(lam >= 0.0) & (lam <= 0.1775)]
    r = fk.feast_hip_hermitian(engine, A, B, 0.0, 0.1775, 64, fpm_with(f2=16, f4=40, f18=aspect), solver="cocg", warm_start=True,
                               inner_rtol=3e-2, solver_maxiter=cap, node_assignment="balanced", column_groups="auto",
                               real_projection=True)
    _cfg3_check(r, A, B, inside)
    # sum mode really ran: the per-loop iteration log exists and no node needed more than the cap
    assert all(max(per_loop) <= cap for per_loop in r.stats["node_iterations"])
    G = r.q.conj().T @ (B @ r.q)
    d = np.sqrt(np.abs(np.diag(G)))
    assert np.abs(G / np.outer(d, d) - np.eye(44)).max() <= 1e-8


def test_cfg3_full_size_properties(engine):
    """BASELINE cfg 3 at full size (N=50 000, nnz=341 500, 16 nodes, M0=64): closed-form spectrum,
    residual recomputed on the host, B-orthonormality of the Ritz vectors."""
    A, B, lam = fo.cfg3_problem(50, 40, 25)
    assert A.shape[0] == 50000 and A.nnz == 341500
    Emin, Emax = 0.0, 0.1775
    inside = lam[(lam >= Emin) & (lam <= Emax)]
    assert len(inside) == 44
    r = fk.feast(A, B, (Emin, Emax), M0=64, fpm=fpm_with(f2=16, f4=40), engine=engine, solver="bicgstab",
                 warm_start=True, inner_rtol=1e-2, solver_maxiter=100, inner_precision=32)
    assert r.info == 0 and r.M == 44 and r.epsout <= 1e-12
    assert np.abs(np.sort(r.lambda_) - inside).max() <= 1e-10
    res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
    assert res.max() <= 1e-10
    G = r.q.T @ (B @ r.q)
    d = np.sqrt(np.diag(G))
    assert np.abs(G / np.outer(d, d) - np.eye(44)).max() <= 1e-8


def test_cfg2_full_size_properties(engine):
    """BASELINE cfg 2 at full size (N=4096 dense real symmetric, 8 nodes, M0=32, batched LU): spectrum known by
    construction (Householder reflection of a diagonal), residual and orthonormality recomputed on the host."""
    N = 4096
    A = fk.workloads.reflected_diagonal(0.01 * np.arange(N))
    lo = 0.01 * (N // 4) - 0.005
    want = 0.01 * np.arange(N // 4, N // 4 + 20)
    r = fk.feast(A, None, (lo, lo + 0.2), M0=32, fpm=fpm_with(f2=8), engine=engine)
    assert r.info == 0 and r.M == 20 and r.epsout <= 1e-12
    assert np.abs(np.sort(r.lambda_) - want).max() <= 1e-10
    assert not np.iscomplexobj(r.q)                       # real input -> real vectors, as feast_sygv! returns
    res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
    assert res.max() <= 1e-10
    assert np.abs(r.q.T @ r.q - np.eye(20)).max() <= 1e-10


@pytest.mark.parametrize("prec", [64, 32])
def test_cfg5_full_size_properties(engine, prec):
    """BASELINE cfg 5 at full size (N=8192 complex non-Hermitian, circle centre 0 radius 2, 24 nodes, M0=48; prec 32 =
    "ComplexF32 mixed precision": complex64 LU + fp64 refinement): eigenvalues known by construction, residual of
    every returned pair recomputed on the host in fp64."""
    A, delta = fk.workloads.disc_spectrum_general(8192)
    A = np.asfortranarray(A)
    inside = delta[np.abs(delta) <= 2.0]
    r = fk.feast_general(A, None, 0.0, 2.0, M0=48, fpm=fpm_with(f8=24, f4=20), engine=engine, inner_precision=prec)
    assert r.info == 0 and r.M == len(inside) == 28
    key = lambda x: (round(x.real, 7), round(x.imag, 7))
    assert np.abs(np.array(sorted(r.lambda_, key=key)) - np.array(sorted(inside, key=key))).max() <= 1e-10
    assert r.epsout <= 1e-11
    res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.linalg.norm(r.q, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
    assert res.max() <= 1e-10
    engine.set_solver("direct")           # back to fp64 factors for the tests that follow


def test_cfg5_reduced_general_dense(engine):
    """cfg 5 shape at N=400: non-normal complex matrix with known eigenvalues in the disc."""
    N = 400
    rng = np.random.default_rng(20260515)
    rad = 6.0 * np.sqrt(rng.random(N))
    delta = rad * np.exp(2j * np.pi * rng.random(N))
    U = np.triu(rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N)), 1) / np.sqrt(N)
    T = np.diag(delta) + 0.05 * U
    v = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    v /= np.linalg.norm(v)
    H = np.eye(N) - 2 * np.outer(v, v.conj())
    A = H @ T @ H
    inside = delta[np.abs(delta) <= 1.5]
    M0 = len(inside) + 14
    r = fk.feast_general(A, None, 0.0, 1.5, M0=M0, fpm=fpm_with(f8=24, f4=40), engine=engine)
    o = fo.feast_general(A, None, 0.0, 1.5, M0, ne=24, fpm4=40)
    assert r.info == 0 and o.info == 0 and r.M == o.M == len(inside)
    key = lambda x: (round(x.real, 8), round(x.imag, 8))
    assert np.allclose(sorted(r.lambda_, key=key), sorted(inside, key=key), atol=1e-9)
    assert np.allclose(sorted(r.lambda_, key=key), sorted(o.lam, key=key), atol=1e-9)
    assert r.epsout <= 1e-11
    # BASELINE config 5 as written: ComplexF32 factors, fp64 refinement -- the same eigenvalues and loop count
    rm = fk.feast_general(A, None, 0.0, 1.5, M0=M0, fpm=fpm_with(f8=24, f4=40), engine=engine, inner_precision=32)
    assert rm.info == 0 and rm.M == len(inside) and abs(rm.loop - r.loop) <= 1
    assert np.allclose(sorted(rm.lambda_, key=key), sorted(inside, key=key), atol=1e-9)
    assert rm.epsout <= 1e-11


def test_empty_and_edge_inputs(engine):
    A = tridiag(20)
    # interval with no eigenvalue: reference reports info=5 / M=0
    r = fk.feast(A, None, (5.0, 6.0), M0=4, engine=engine)
    o = fo.feast_hermitian(A, None, 5.0, 6.0, 4)
    assert r.M == 0 == o.M and r.info == 5 == o.info
    # M0 == N (maximum) and M0 clipped to N like feast() does
    r = fk.feast(A, None, (0.0, 4.0), M0=64, engine=engine)
    assert r.info == 0 and r.M == 20
    # singular shift on the contour cannot happen for Hermitian input (Im z != 0); a 1x1 problem works
    r = fk.feast(np.array([[2.0]]), None, (1.0, 3.0), M0=1, engine=engine)
    assert r.info == 0 and r.M == 1 and abs(r.lambda_[0] - 2.0) < 1e-12


def test_contour_variants_trapezoid_and_ellipse(engine):
    """fpm[16]=1 (trapezoid) and fpm[18]=50 (flat ellipse) only change the host-side nodes and
    weights (src/core/feast_tools.jl:212-284); the device sweep must follow."""
    N = 300
    A = fo.householder_conjugated_diag(0.02 * np.arange(N))
    want = 0.02 * np.arange(100, 108)
    for kw in (dict(f16=1, f2=12), dict(f18=50, f2=8), dict(f16=1, f18=30, f2=16), dict(f16=2, f2=8, f4=40)):   # f16=2: Zolotarev
        r = fk.feast(A, None, (1.99, 2.15), M0=20, fpm=fpm_with(**kw), engine=engine)
        assert r.info == 0 and r.M == 8 and np.allclose(np.sort(r.lambda_), want, atol=1e-10), kw
        o = fo.feast_hermitian(A, None, 1.99, 2.15, 20, ne=kw["f2"], fpm16=kw.get("f16", 0), fpm18=kw.get("f18", 100),
                               fpm4=kw.get("f4", 20), real_projection=True)
        assert o.info == 0 and np.allclose(np.sort(o.lam), want, atol=1e-10)


def test_complex_hermitian_sparse_generalized_bicgstab(engine):
    """Complex Hermitian A, B (no real projection, half contour): BiCGStab path, parity with the
    oracle's direct variant A and with dense eigh."""
    N = 400
    rng = np.random.default_rng(7)
    S = sp.random(N, N, density=4.0 / N, random_state=3, format="csr")
    S = S + 1j * sp.random(N, N, density=4.0 / N, random_state=4, format="csr")
    A = sp.csr_matrix(S + S.conj().T + sp.diags(np.linspace(1.0, 40.0, N)))
    T = sp.random(N, N, density=2.0 / N, random_state=5, format="csr") * (0.3 + 0.2j)
    B = sp.csr_matrix(T + T.conj().T + sp.diags(2.0 + rng.random(N)))
    import scipy.linalg as sla
    ev = sla.eigh(A.toarray(), B.toarray(), eigvals_only=True)
    lo, hi = ev[5] - 1e-3, ev[14] + 1e-3
    inside = ev[(ev >= lo) & (ev <= hi)]
    r = fk.feast(A, B, (lo, hi), M0=len(inside) + 14, fpm=fpm_with(f2=8, f4=80), engine=engine, solver="bicgstab",
                 solver_tol=1e-13, solver_maxiter=3000)
    o = fo.feast_hermitian(A, B, lo, hi, len(inside) + 14, ne=8, fpm4=80)
    assert r.M == len(inside) == o.M and r.info == o.info == 0
    assert np.allclose(np.sort(r.lambda_), inside, atol=1e-9) and np.allclose(np.sort(o.lam), inside, atol=1e-9)
    res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
    assert res.max() <= 1e-10
    with pytest.raises(fk.FeastHipError):       # COCG is only for complex-symmetric shifted systems
        fk.feast(A, B, (lo, hi), M0=len(inside) + 14, engine=engine, solver="cocg", warm_start=True, inner_rtol=1e-2)


def test_dense_generalized_real_symmetric_cfg2_variant(engine):
    """cfg 2 variant with B = H diag(1 + 0.5 u) H, N = 512."""
    N = 512
    rng = np.random.default_rng(20260515)
    A = fo.householder_conjugated_diag(0.01 * np.arange(N))
    v = rng.standard_normal(N); v /= np.linalg.norm(v)
    H = np.eye(N) - 2 * np.outer(v, v)
    B = H @ np.diag(1 + 0.5 * rng.random(N)) @ H
    B = 0.5 * (B + B.T)
    import scipy.linalg as sla
    ev = sla.eigh(A, B, eigvals_only=True)
    lo, hi = 0.5 * (ev[99] + ev[100]), 0.5 * (ev[118] + ev[119])     # edges in the middle of spectral gaps
    inside = ev[(ev >= lo) & (ev <= hi)]
    r = fk.feast(A, B, (lo, hi), M0=32, fpm=fpm_with(f2=8), engine=engine)
    assert r.info == 0 and r.M == len(inside) and np.allclose(np.sort(r.lambda_), inside, atol=1e-10)
    assert r.epsout <= 1e-12


def test_variant_b_moments_on_gpu(engine):
    """Variant B through the C ABI (moments zAq/zSq from feasthip_contour_apply), the PFEAST fixture
    of runtests.jl:1042-1089, dense (LU) and sparse (BiCGStab)."""
    k = K["diag4_variant_b"]
    A = np.diag(k["diag"]); B = np.eye(4)
    for Ain, Bin in ((A, B), (sp.csr_matrix(A), sp.identity(4, format="csr"))):
        r = fk.pfeast_hip_moments(engine, Ain, Bin, *k["interval"], 4, fpm_with(f2=k["fpm2"], f4=k["fpm4"]))
        o = fo.pfeast_moments(sp.csc_matrix(Ain) if sp.issparse(Ain) else Ain, sp.csc_matrix(Bin) if sp.issparse(Bin) else Bin,
                              *k["interval"], 4, ne=k["fpm2"], fpm4=k["fpm4"])
        assert r.info == 0 == o.info and r.M == 3 == o.M
        assert np.allclose(r.lambda_, k["expect_lambda"], atol=k["atol"]) and np.allclose(r.lambda_, o.lam, atol=1e-9)


def test_hermitian_moments_driver_on_gpu(engine):
    """Variant B for complex Hermitian input through the C ABI (the image of _mpi_feast_complex_hermitian!,
    src/parallel/feast_mpi.jl:796-909): reference fixture, dense LU and sparse BiCGStab, against the oracle."""
    k = K["mpi_complex_hermitian_diag4"]
    Ad = np.diag(np.array(k["diag"], dtype=complex)); Bd = np.eye(4, dtype=complex)
    want = fo.mpi_complex_hermitian(Ad, Bd, *k["interval"], 4, ne=k["fpm2"], fpm4=k["fpm4"])
    r = fk.pfeast_hip_hermitian_moments(engine, Ad, Bd, *k["interval"], 4, fpm_with(f2=k["fpm2"], f4=k["fpm4"]))
    assert (r.info, r.M, r.loop) == (0, 3, want.loop) == (want.info, want.M, want.loop)
    assert np.allclose(r.lambda_, k["expect_lambda"], atol=k["atol"]) and np.allclose(r.lambda_, want.lam, atol=1e-12)
    # sparse input: no sparse LU on the device, the shifted systems go through the device GMRES.  This un-normalised
    # iteration amplifies the solver's error (about 50x in the first loop, 3x per further loop -- measured), so the
    # Krylov path is asked for what it can deliver: outer tolerance 1e-9 (the fixture's own bar is atol 1e-8)
    r = fk.pfeast_hip_hermitian_moments(engine, sp.csr_matrix(Ad), sp.identity(4, dtype=complex, format="csr"), *k["interval"], 4,
                                        fpm_with(f2=k["fpm2"], f3=9, f4=k["fpm4"]))
    assert (r.info, r.M) == (0, 3) and r.epsout <= 1e-9
    assert np.allclose(r.lambda_, k["expect_lambda"], atol=k["atol"])
    # genuinely complex Hermitian pencil: loop-for-loop agreement with the oracle
    n = 40
    rng = np.random.default_rng(12)
    H = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = np.diag(np.linspace(0.0, 8.0, n)) + 0.05 * (H + H.conj().T)
    ev = np.linalg.eigvalsh(A)
    lo, hi = 0.5 * (ev[6] + ev[7]), 0.5 * (ev[13] + ev[14])
    for loops in (2, 6):
        r = fk.pfeast_hip_hermitian_moments(engine, A, np.eye(n, dtype=complex), lo, hi, 12, fpm_with(f2=8, f3=11, f4=loops))
        o = fo.mpi_complex_hermitian(A, None, lo, hi, 12, ne=8, fpm3=11, fpm4=loops)
        assert (r.info, r.M, r.loop) == (o.info, o.M, o.loop)
        assert np.allclose(r.lambda_, o.lam, atol=1e-8)


# ---- complex-symmetric sibling of variant A (src/dense/feast_dense.jl:1026-1259, feast_sparse.jl:509-711)
def _complex_symmetric_problem(n, seed=7, generalized=False):
    rng = np.random.default_rng(seed)
    d = np.linspace(-2.0, 2.0, n) + 1j * rng.uniform(-0.6, 0.6, n)
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))          # real orthogonal: A = Qo D Qo^T is complex symmetric
    A = Qo @ np.diag(d) @ Qo.T
    A = 0.5 * (A + A.T)
    B = None
    if generalized:
        B = Qo @ np.diag(1.0 + 0.3 * rng.random(n)) @ Qo.T
        B = (0.5 * (B + B.T)).astype(complex)
    return A, B, d


@pytest.mark.parametrize("generalized", [False, True])
def test_complex_symmetric_dense_matches_oracle(engine, generalized):
    A, B, _ = _complex_symmetric_problem(48, generalized=generalized)
    ev = np.linalg.eigvals(A if B is None else np.linalg.solve(B, A))
    c = 0.2 + 0.05j
    dist = np.sort(np.abs(ev - c))
    r = 0.5 * (dist[7] + dist[8])
    want = fo.feast_complex_symmetric(A, B, c, r, 12, ne=16, fpm3=11, fpm4=20)
    fpm = fk.feastinit(); fpm[8] = 16; fpm[3] = 11; fpm[4] = 20
    got = fk.feast_hip_complex_symmetric(engine, A, B, c, r, 12, fpm)
    assert (got.info, got.M) == (want.info, want.M) == (0, 8)
    assert abs(got.loop - want.loop) <= 1
    inside = ev[np.abs(ev - c) <= r]
    key = lambda x: (round(x.real, 6), round(x.imag, 6))
    assert np.allclose(sorted(got.lambda_, key=key), sorted(inside, key=key), atol=1e-9)
    assert np.allclose(sorted(got.lambda_, key=key), sorted(want.lam, key=key), atol=1e-9)
    assert got.epsout <= 1e-11
    X = got.q
    Bm = np.eye(48) if B is None else B
    assert np.linalg.norm(A @ X - (Bm @ X) * got.lambda_[None, :]) < 1e-9


def test_complex_symmetric_sparse_bicgstab(engine):
    """CSR complex-symmetric pencil: tridiagonal with complex diagonal (a cluster of 10 entries near 2.7,
    the rest in [4.5, 8]), Krylov solves on the device."""
    n = 300
    rng = np.random.default_rng(2)
    diag = np.linspace(4.5, 8.0, n) + 0.3j * rng.uniform(-1, 1, n)
    idx = np.arange(10) * 29 + 5
    diag[idx] = 2.7 + 0.15 * rng.uniform(-1, 1, 10) + 0.1j * rng.uniform(-1, 1, 10)
    A = sp.diags([-0.2 * np.ones(n - 1), diag, -0.2 * np.ones(n - 1)], [-1, 0, 1], format="csr").astype(complex)
    ev = np.linalg.eigvals(A.toarray())
    c, r = 2.7 + 0.0j, 0.9
    inside = ev[np.abs(ev - c) <= r]
    fpm = fk.feastinit(); fpm[8] = 16; fpm[3] = 10; fpm[4] = 25
    got = fk.feast_hip_complex_symmetric(engine, A, None, c, r, 16, fpm, solver="bicgstab", solver_tol=1e-12, solver_maxiter=3000)
    assert got.info == 0 and got.M == len(inside) == 10
    key = lambda x: (round(x.real, 6), round(x.imag, 6))
    assert np.allclose(sorted(got.lambda_, key=key), sorted(inside, key=key), atol=1e-8)
    assert got.epsout <= 1e-10


# ---- M0 > 64: every entry point works panel by panel (64 columns) ---------------------------------
def test_wide_subspace_dense_and_sparse(engine):
    """M0 > 64.  Dense LU path: 80 clustered eigenvalues + a far-away rest (the 100-column subspace is
    compressed to rank ~80), then a uniformly dense spectrum where the reference algorithm stagnates
    on spurious Ritz values -- the GPU path must stagnate identically.  CSR + COCG: ~75 wanted."""
    N = 500
    lo, hi = 1.005, 1.805
    want = 0.01 * np.arange(101, 181)
    A = fo.householder_conjugated_diag(np.concatenate([want, np.linspace(3.0, 10.0, N - 80)]))
    r = fk.feast(A, None, (lo, hi), M0=100, fpm=fpm_with(f2=8), engine=engine)
    assert r.info == 0 and r.M == 80 and np.allclose(np.sort(r.lambda_), want, atol=1e-10)
    res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
    assert res.max() <= 1e-10
    o = fo.feast_hermitian(A, None, lo, hi, 100, ne=8, real_projection=True)
    assert o.info == 0 and o.M == 80 and abs(o.loop - r.loop) <= 1
    A2 = fo.householder_conjugated_diag(0.01 * np.arange(N))
    r8 = fk.feast(A2, None, (lo, hi), M0=100, fpm=fpm_with(f2=8, f4=4), engine=engine)
    o8 = fo.feast_hermitian(A2, None, lo, hi, 100, ne=8, fpm4=4, real_projection=True)
    assert (r8.info, r8.M, r8.loop) == (o8.info, o8.M, o8.loop) == (5, 81, 4)
    assert abs(r8.epsout - o8.epsout) <= 1e-6 * o8.epsout
    # sparse generalized, warm-started inexact COCG, Ritz warm start sliced per panel
    As, Bs, lam = fo.cfg3_problem(14, 12, 10)
    hi = 0.5 * (lam[74] + lam[75])
    inside = lam[:75]
    rs = fk.feast_hip_hermitian(engine, As, Bs, 0.0, hi, 95, fpm_with(f2=8, f4=40), solver="cocg", warm_start=True,
                                inner_rtol=1e-2, solver_maxiter=300, real_projection=True)
    assert rs.info == 0 and rs.M == 75 and np.allclose(np.sort(rs.lambda_), inside, atol=1e-10)
    assert rs.epsout <= 1e-12


def test_wide_general_dense(engine):
    """Variant C with M0 = 80 (> 64) on a non-Hermitian dense matrix."""
    n = 300
    rng = np.random.default_rng(12)
    d = np.concatenate([0.9 * np.sqrt(rng.random(70)) * np.exp(2j * np.pi * rng.random(70)),
                        (2.0 + 3.0 * rng.random(n - 70)) * np.exp(2j * np.pi * rng.random(n - 70))])
    S = rng.standard_normal((n, n)) + n * np.eye(n)
    A = S @ np.diag(d) @ np.linalg.inv(S)
    r = fk.feast_general(A, None, 0.0 + 0.0j, 1.2, M0=80, fpm=fpm_with(f8=16), engine=engine)
    inside = d[np.abs(d) <= 1.2]
    assert r.info == 0 and r.M == len(inside) == 70
    key = lambda x: (round(x.real, 6), round(x.imag, 6))
    assert np.allclose(sorted(r.lambda_, key=key), sorted(inside, key=key), atol=1e-8)


def test_single_precision_element_types_preserved(engine):
    """test/runtests.jl:281-304: Float32 / ComplexF32 input runs and keeps its element type in lambda and q (the :hip shim
    promotes on the way in -- the C ABI is f64 / c128 -- and stops at max(10^-fpm[3], sqrt(eps(Float32))) like
    src/core/feast_parameters.jl:398-405)."""
    n = 4
    A = (np.diag(np.full(n, 2.0)) - np.diag(np.ones(n - 1), 1) - np.diag(np.ones(n - 1), -1)).astype(np.float32)
    B = np.eye(n, dtype=np.float32)
    r = fk.feast(A, B, (np.float32(0.0), np.float32(4.0)), M0=n, engine=engine)
    assert r.info == 0 and r.M >= 1 and r.lambda_.dtype == np.float32 and r.q.dtype == np.float32 and r.res.dtype == np.float32
    want = 2.0 - 2.0 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))
    assert r.M == n and np.abs(np.sort(r.lambda_) - want).max() < 1e-5
    Ac = np.diag(np.array([0.25, 1.25, 2.25, 3.25], dtype=np.complex64))
    rc = fk.feast(Ac, None, (-2.0, 2.0), M0=n, engine=engine)
    assert rc.info == 0 and rc.M == 2 and rc.q.dtype == np.complex64 and rc.lambda_.dtype == np.float32
    assert np.abs(np.sort(rc.lambda_) - [0.25, 1.25]).max() < 1e-5
    As = sp.csr_matrix(A)
    rs = fk.feast(As, None, (np.float32(0.0), np.float32(4.0)), M0=n, engine=engine)
    assert rs.info == 0 and rs.M == n and rs.lambda_.dtype == np.float32 and rs.q.dtype == np.float32
    G = np.array([[1, 2 + 1j], [0, 3]], dtype=np.complex64)
    rg = fk.feast_general(G, None, 2.0, 2.5, M0=2, engine=engine)
    assert rg.info == 0 and rg.M == 2 and rg.lambda_.dtype == np.complex64 and rg.q.dtype == np.complex64
    assert np.abs(np.sort(rg.lambda_.real) - [1.0, 3.0]).max() < 1e-5
    # double-precision input is untouched
    rd = fk.feast(A.astype(np.float64), None, (0.0, 4.0), M0=n, engine=engine)
    assert rd.lambda_.dtype == np.float64 and rd.q.dtype == np.float64 and rd.epsout <= 1e-12
