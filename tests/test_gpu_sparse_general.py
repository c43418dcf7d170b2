"""The sparse GENERAL driver (feast_gcsrgv!/feast_gcsrev!, src/sparse/feast_sparse.jl:873-1006) and the
caller-supplied-contour ("x") drivers on the :hip backend, through the C ABI on the GPU.

Fixtures are the reference's own: the sparse 2x2 of test/runtests.jl:225-238, the direct == GMRES general
pencil of :487-510, the MPI complex-general fixture of test/test_parallel_backends.jl:125-141, the
Hermitian generalized "x" fixture of test/runtests.jl:415-432 -- plus a cfg-5-shaped sparse non-normal
pencil against the oracle's feast_general (variant C, src/kernel/feast_kernel.jl:752-950)."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import feast_oracle as fo
import feastkit_jl_amd as fk
from kat_util import cmat, cplx, load_kats

pytestmark = pytest.mark.gpu
K = load_kats()
ckey = lambda x: (round(x.real, 8), round(x.imag, 8))


def fpm_with(**kw):
    fpm = fk.feastinit()
    for k, v in kw.items():
        fpm[int(k[1:])] = v
    return fpm


def test_general2_sparse_default_call(engine):
    """runtests.jl:225-231: feast_general(sparse(A), center, radius; M0) with every keyword at its default.
    The reference factors with UMFPACK; here the default maps to the banded LU (the 2x2 pattern is a band)."""
    k = K["general2"]
    A = sp.csr_matrix(cmat(k["A"]))
    r = fk.feast_general(A, None, cplx(k["center"]), k["radius"], M0=2, engine=engine)
    assert r.info == 0 and r.M == 2
    assert np.allclose(np.sort(r.lambda_.real), k["expect_standard"], atol=k["atol"])
    B = sp.csr_matrix(cmat(k["B"]))
    r = fk.feast_general(A, B, cplx(k["center"]), k["radius"], M0=2, engine=engine)
    assert r.info == 0 and r.M == 2
    assert np.allclose(np.sort(r.lambda_.real), k["expect_generalized"], atol=k["atol"])


def _corner_coupled_diag(delta):
    """diag(delta) + a sparse superdiagonal + two far-corner entries: non-normal, and NOT a narrow band."""
    N = len(delta)
    A = sp.lil_matrix(sp.diags(delta), dtype=np.complex128)
    for i in range(0, N - 1, 7):
        A[i, i + 1] = 0.02
    A[0, N - 1] = 0.01 + 0.02j
    A[N - 1, 3] = -0.015j
    return sp.csr_matrix(A)


def test_sparse_general_default_call_wide_pattern(engine):
    """The call that raised ValueError in round 2: feast_general(A_csr) with default keywords on a pattern that is not
    a narrow band.  At this size `solver=:direct` (UMFPACK in the reference) is served by the batched dense LU on the
    expanded matrix -- an exact direct solve.  The spectrum SURROUNDS the contour here (a disc of radius 4 around a
    contour of radius 0.8), the situation of BASELINE cfg 5, where no unpreconditioned Krylov method converges (a
    polynomial with p(0) = 1 cannot be small on a disc around 0): the reference's own GMRES option returns info = 5."""
    N = 2400
    rng = np.random.default_rng(11)
    delta = 4.0 * np.sqrt(rng.random(N)) * np.exp(2j * np.pi * rng.random(N))
    delta[:6] = [0.3 + 0.1j, -0.2 + 0.4j, 0.5 - 0.3j, -0.4 - 0.2j, 0.1 + 0.6j, 0.0 - 0.5j]
    delta[6:] = np.where(np.abs(delta[6:]) < 1.2, delta[6:] * 1.2 / np.maximum(np.abs(delta[6:]), 1e-3), delta[6:])
    A = _corner_coupled_diag(delta)
    assert fk.api._sparse_direct_solver(A, None, 16) == "dense"
    # M0 = the number of eigenvalues inside: variant C carries all M0 columns un-orthonormalised, and columns beyond the
    # invariant subspace are filtered to rounding noise within a few loops (the reduced B matrix turns singular and
    # spurious Ritz values appear -- the oracle shows the same with M0 = 7; the reference's own general tests all use
    # M0 = n for that reason)
    r = fk.feast_general(A, None, 0.0, 0.8, M0=6, fpm=fpm_with(f8=16), engine=engine)
    assert r.stats["solver_substitution"]["used"].startswith("dense LU")
    assert r.info == 0 and r.M == 6
    ev = np.linalg.eigvals(A.toarray())
    want = ev[np.abs(ev) <= 0.8]
    assert np.allclose(sorted(r.lambda_, key=ckey), sorted(want, key=ckey), atol=1e-9)
    o = fo.feast_general(A, None, 0.0, 0.8, 6, ne=16)
    assert o.info == 0 and o.M == 6 and np.allclose(sorted(o.lam, key=ckey), sorted(r.lambda_, key=ckey), atol=1e-9)
    assert r.loop == o.loop
    res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.linalg.norm(r.q, axis=0)
    assert res.max() <= 1e-9
    # the same matrix through the iterative keyword: the reference semantics are "return info = 5", not an exception
    ri = fk.feast_general(A, None, 0.0, 0.8, M0=6, fpm=fpm_with(f8=16), engine=engine, solver="gmres", solver_maxiter=60)
    assert ri.info == 5 and ri.M == 0


def _arrow_coupled_diag(delta, eps=1e-3):
    """diag(delta) + a sparse superdiagonal + a full first row and column (a star graph: no renumbering gives it a band
    narrower than N / 2), which a fill-reducing sparse LU factors without fill."""
    N = len(delta)
    j = np.arange(1, N)
    rows = np.concatenate([np.arange(N), np.zeros(N - 1, dtype=int), j, np.arange(0, N - 1, 7)])
    cols = np.concatenate([np.arange(N), j, np.zeros(N - 1, dtype=int), np.arange(1, N, 7)])
    vals = np.concatenate([delta, np.full(N - 1, eps * (1 + 2j)), np.full(N - 1, -1.5j * eps), np.full(len(range(0, N - 1, 7)), 0.02)])
    return sp.csr_matrix(sp.coo_matrix((vals, (rows, cols)), shape=(N, N)))


def _clustered_spectrum(N, seed=12):
    rng = np.random.default_rng(seed)
    delta = 10.0 + 2.0 * np.sqrt(rng.random(N)) * np.exp(2j * np.pi * rng.random(N))
    delta[1:7] = [0.3 + 0.1j, -0.2 + 0.4j, 0.5 - 0.3j, -0.4 - 0.2j, 0.1 + 0.6j, 0.0 - 0.5j]
    return delta


def test_sparse_general_default_call_large_maps_to_band_direct(engine):
    """Beyond the dense window the default `solver=:direct` (UMFPACK in the reference) is served by the sparse direct solver
    for general patterns whenever its factors fit the device: reverse Cuthill-McKee + blocked band LU.  The two far-corner
    entries make the stored band N wide; the renumbering finds the narrow one."""
    N = 14000
    A = _corner_coupled_diag(_clustered_spectrum(N))
    assert fk.api._sparse_direct_solver(A, None, 16) == "krylov"             # not a narrow band as stored, too large for dense
    r = fk.feast_general(A, None, 0.0, 0.8, M0=6, fpm=fpm_with(f8=16), engine=engine)
    assert r.stats["solver_substitution"]["used"].split()[0] in ("band", "multifrontal")
    assert r.info == 0 and r.M == 6 and r.stats["krylov_iterations"] == 0 and r.stats["factorizations"] == 16
    kl, ku, nbytes, blocked = engine.band_plan()
    assert kl + ku <= 64                    # (the ingest renumbering or the band plan's own: either way a narrow band again)
    o = fo.feast_general(A, None, 0.0, 0.8, 6, ne=16)                     # sparse LU per node
    assert o.info == 0 and o.M == 6
    assert np.allclose(sorted(r.lambda_, key=ckey), sorted(o.lam, key=ckey), atol=1e-10)
    assert r.loop == o.loop and r.epsout <= 1e-11
    res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.linalg.norm(r.q, axis=0)
    assert res.max() <= 1e-10


def test_sparse_general_default_call_large_maps_to_krylov(engine, monkeypatch):
    """A pattern no renumbering can make a band of (a full row and column) at N = 14 000: beyond the dense window and the
    BAND solver's reach (FH_MF=0: no multifrontal plan), the default maps to batched BiCGStab with the reference's iterative
    settings (zero guess, rtol = atol = 10^-fpm[3], 500 iterations), recorded in stats and warned once.  The bulk of the
    spectrum lies in a disc AWAY from the contour (centre 10, radius 2) with six eigenvalues inside it: z - A then has a
    clustered spectrum off the origin plus six outliers and BiCGStab converges in a few dozen iterations."""
    monkeypatch.setenv("FH_MF", "0")
    N = 14000
    A = _arrow_coupled_diag(_clustered_spectrum(N))
    assert fk.api._sparse_direct_solver(A, None, 16) == "krylov"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        fk.api._warned.clear()
        r = fk.feast_general(A, None, 0.0, 0.8, M0=6, fpm=fpm_with(f8=16), engine=engine)
    assert any("instead of a sparse LU" in str(x.message) for x in w)
    assert r.stats["solver_substitution"]["used"] == "bicgstab"
    assert r.info == 0 and r.M == 6 and r.stats["krylov_iterations"] > 0
    o = fo.feast_general(A, None, 0.0, 0.8, 6, ne=16)                     # sparse LU per node
    assert o.info == 0 and o.M == 6
    assert np.allclose(sorted(r.lambda_, key=ckey), sorted(o.lam, key=ckey), atol=1e-9)
    assert abs(r.loop - o.loop) <= 1 and r.epsout <= 1e-11
    res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.linalg.norm(r.q, axis=0)
    assert res.max() <= 1e-10
    rg = fk.feast_general(A, None, 0.0, 0.8, M0=6, fpm=fpm_with(f8=16), engine=engine, solver="gmres")
    assert rg.info == 0 and np.allclose(sorted(rg.lambda_, key=ckey), sorted(o.lam, key=ckey), atol=1e-9)


def test_sparse_general_arrow_pattern_takes_the_multifrontal_solver(engine):
    """The same kind of arrow pattern with the library's own choice: no band, but a one-vertex separator -- the multifrontal
    plan eliminates it like the reference's UMFPACK call would, and the default `solver=:direct` is a direct solve again
    (same loops as the oracle's sparse LU per node, no Krylov iterations)."""
    N = 14000
    A = _arrow_coupled_diag(_clustered_spectrum(N, seed=13))      # (another matrix: the engine keeps the plan of one it holds)
    r = fk.feast_general(A, None, 0.0, 0.8, M0=6, fpm=fpm_with(f8=16), engine=engine)
    assert r.stats["solver_substitution"]["used"].startswith("multifrontal")
    assert r.info == 0 and r.M == 6 and r.stats["krylov_iterations"] == 0 and r.stats["factorizations"] == 16
    o = fo.feast_general(A, None, 0.0, 0.8, 6, ne=16)
    assert o.info == 0 and o.M == 6
    assert np.allclose(sorted(r.lambda_, key=ckey), sorted(o.lam, key=ckey), atol=1e-10)
    assert r.loop == o.loop and r.epsout <= 1e-11
    res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.linalg.norm(r.q, axis=0)
    assert res.max() <= 1e-10


@pytest.mark.parametrize("generalized", [True, False])
def test_sparse_general_direct_equals_gmres(engine, generalized):
    """runtests.jl:482-510: feast_gcsrgv!/feast_gcsrev! with solver=:direct and with
    solver=:gmres, solver_tol=1e-7, solver_maxiter=400, solver_restart=30 agree to 1e-7 / 1e-6."""
    d = np.array([1.0 + 0.1j, 1.5 - 0.2j, 2.0 + 0.3j, 2.8 - 0.1j, 3.5 + 0.2j, 4.5])
    n = len(d)
    A = sp.diags(d).tocsr()
    B = sp.identity(n, dtype=np.complex128, format="csr") if generalized else None
    direct = fk.feast_general(A, B, 2.0 + 0.0j, 3.0, M0=n, engine=engine)
    gm = fk.feast_general(A, B, 2.0 + 0.0j, 3.0, M0=n, engine=engine, solver="gmres", solver_tol=1e-7,
                          solver_maxiter=400, solver_restart=30)
    assert direct.info == 0 and gm.info == 0 and gm.M == direct.M == n
    assert np.allclose(np.sort(gm.lambda_.real), np.sort(direct.lambda_.real), atol=1e-7)
    assert np.allclose(sorted(direct.lambda_, key=ckey), sorted(d, key=ckey), atol=1e-9)
    it = fk.feast_general(A, B, 2.0 + 0.0j, 3.0, M0=n, engine=engine, solver="iterative", solver_tol=1e-7,
                          solver_maxiter=400)
    assert it.info == 0 and np.allclose(np.sort(it.lambda_.real), np.sort(direct.lambda_.real), atol=1e-6)


def test_mpi_complex_general_fixture_on_csr(engine):
    """test/test_parallel_backends.jl:125-141: diag(.5+.1i, 1+.2i, 2-.1i, 4), centre 1+.1i, r 1.3,
    fpm[3]=11, fpm[4]=12, fpm[8]=12 -> the three eigenvalues inside, atol 1e-8; CSR input, B = I given explicitly."""
    k = K["mpi_complex_general_diag4"]
    d = np.array([cplx(v) for v in k["diag"]])
    A = sp.diags(d).tocsr()
    B = sp.identity(len(d), dtype=np.complex128, format="csr")
    want = [cplx(v) for v in k["expect_lambda"]]
    fp = dict(f3=k["fpm3"], f4=k["fpm4"], f8=k["fpm8"])
    for solver in ("direct", "bicgstab", "gmres"):
        r = fk.feast_general(A, B, cplx(k["center"]), k["radius"], M0=len(d), fpm=fpm_with(**fp), engine=engine,
                             solver=solver, solver_maxiter=400)
        assert r.info == 0 and r.M == len(want), (solver, r.info, r.M)
        assert np.allclose(sorted(r.lambda_, key=ckey), sorted(want, key=ckey), atol=k["atol"]), solver
    o = fo.feast_general(A, B, cplx(k["center"]), k["radius"], len(d), ne=k["fpm8"], fpm3=k["fpm3"], fpm4=k["fpm4"])
    assert o.M == len(want) and np.allclose(sorted(o.lam, key=ckey), sorted(want, key=ckey), atol=k["atol"])


def test_cfg5_shaped_sparse_non_normal_vs_oracle(engine):
    """cfg 5 in sparse clothing, reduced: T = diag(delta) + 0.05 U with U sparse strictly upper (non-normal), a
    sparse similarity by 2x2 rotations so the matrix is not triangular; eigenvalues = delta exactly.  The default call
    against fo.feast_general (sparse LU per node), loop for loop."""
    N = 1500
    rng = np.random.default_rng(20260515)
    delta = 6.0 * np.sqrt(rng.random(N)) * np.exp(2j * np.pi * rng.random(N))
    U = sp.triu(sp.random(N, N, density=4.0 / N, random_state=np.random.RandomState(5), format="csr"), 1)
    U = U.astype(np.complex128) * (1 + 0.5j) / np.sqrt(8.0)
    T = sp.diags(delta) + 0.05 * U
    # block-diagonal unitary of 2x2 complex rotations on a random pairing: G T G^H is sparse and not triangular
    perm = rng.permutation(N)
    rows, cols, vals = [], [], []
    for a, b in zip(perm[0::2], perm[1::2]):
        th, ph = rng.random() * np.pi, rng.random() * 2 * np.pi
        c, s = np.cos(th), np.sin(th) * np.exp(1j * ph)
        rows += [a, a, b, b]
        cols += [a, b, a, b]
        vals += [c, -np.conj(s), s, c]
    G = sp.csr_matrix((vals, (rows, cols)), shape=(N, N))
    A = sp.csr_matrix(G @ T @ G.conj().T)
    rad = 0.8168                                   # a gap of |delta|: 0.7993 | 0.8343
    inside = delta[np.abs(delta) <= rad]
    M0 = len(inside) + 12
    assert len(inside) == 30
    o = fo.feast_general(A, None, 0.0, rad, M0, ne=24, fpm4=40)
    assert o.info == 0 and o.M == len(inside)
    # default call: solver=:direct.  The spectrum surrounds the contour (cfg 5's situation), so this is a job for a
    # direct solver; the sparse pattern is expanded and factored by the batched dense LU, in complex128 and (BASELINE
    # cfg 5 as written) with complex64 factors + fp64 refinement
    for prec in (64, 32):
        r = fk.feast_general(A, None, 0.0, rad, M0=M0, fpm=fpm_with(f8=24, f4=40), engine=engine, inner_precision=prec)
        assert r.info == 0 and r.M == len(inside), (prec, r.info, r.M)
        assert np.allclose(sorted(r.lambda_, key=ckey), sorted(inside, key=ckey), atol=1e-9), prec
        assert np.allclose(sorted(r.lambda_, key=ckey), sorted(o.lam, key=ckey), atol=1e-9), prec
        assert abs(r.loop - o.loop) <= 1, (prec, r.loop, o.loop)
        assert r.epsout <= 1e-11
        res = np.linalg.norm(A @ r.q - r.q * r.lambda_, axis=0) / np.linalg.norm(r.q, axis=0) / np.maximum(np.abs(r.lambda_), 1)
        assert res.max() <= 1e-10, prec


def test_custom_contour_hermitian_x_driver(engine):
    """runtests.jl:415-432 (feast_hcsrgvx!): nodes and weights handed in by the caller give the same eigenvalues
    as the built-in contour.  Then a contour feast_contour cannot produce (trapezoid nodes on a tall ellipse passed
    as plain arrays) straight through feast_hip_hermitian(contour=...) against the oracle on the same nodes."""
    A = sp.diags(np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], dtype=np.complex128)).tocsr()
    B = sp.diags(np.array([1.0, 1.2, 1.5, 2.5, 4.0, 5.0], dtype=np.complex128)).tocsr()
    want = np.sort([l for l in np.arange(1.0, 7.0) / np.array([1.0, 1.2, 1.5, 2.5, 4.0, 5.0]) if 0.5 <= l <= 3.1])
    fpm = fk.feastinit()
    fk.feastdefault(fpm)
    base = fk.feast(A, B, (0.5, 3.1), M0=6, fpm=fpm.copy(), engine=engine, solver="bicgstab", solver_maxiter=200)
    Zne, Wne = fk.feast_contour(0.5, 3.1, fpm.copy())
    rx = fk.feast(A, B, (0.5, 3.1), M0=6, fpm=fpm.copy(), engine=engine, solver="bicgstab", solver_maxiter=200,
                  contour=(Zne, Wne))
    assert base.info == 0 and rx.info == 0 and rx.M == base.M == len(want)
    assert np.allclose(np.sort(rx.lambda_), np.sort(base.lambda_), atol=1e-8)
    assert np.allclose(np.sort(rx.lambda_), want, atol=1e-8)

    # hand-made half contour: 10 midpoint-rule nodes on the upper half of an ellipse with vertical semi-axis 2.5 r
    N = 300
    Ad = fo.householder_conjugated_diag(0.02 * np.arange(N))
    Emin, Emax = 1.99, 2.15
    mid, rr, asp, ne = 0.5 * (Emin + Emax), 0.5 * (Emax - Emin), 2.5, 10
    th = np.pi * (np.arange(ne) + 0.5) / ne
    Z = mid + rr * np.cos(th) + 1j * rr * asp * np.sin(th)
    W = (rr * asp * np.cos(th) + 1j * rr * np.sin(th)) / (2.0 * ne)       # dz/(2 pi i) per node, half contour
    r = fk.feast_hip_hermitian(engine, Ad, None, Emin, Emax, 20, fk.feastinit(), contour=(Z, W))
    o = fo.feast_hermitian(Ad, None, Emin, Emax, 20, contour=(Z, W), real_projection=True)
    assert r.info == 0 and o.info == 0 and r.M == o.M == 8
    assert np.allclose(np.sort(r.lambda_), 0.02 * np.arange(100, 108), atol=1e-10)
    assert np.allclose(np.sort(r.lambda_), np.sort(o.lam), atol=1e-10)
    assert abs(r.loop - o.loop) <= 1


def test_custom_contour_general_x_driver(engine):
    """feast_gcsrgvx!: a caller-supplied full contour through the sparse general driver."""
    d = np.array([1.0 + 0.1j, 1.5 - 0.2j, 2.0 + 0.3j, 2.8 - 0.1j, 3.5 + 0.2j, 4.5])
    A = sp.diags(d).tocsr()
    fpm = fpm_with(f8=16)
    fk.feastdefault(fpm)
    Zne, Wne = fk.feast_gcontour(2.0 + 0.0j, 1.0, fpm.copy())
    base = fk.feast_general(A, None, 2.0, 1.0, M0=5, fpm=fpm.copy(), engine=engine)
    rx = fk.feast_general(A, None, 2.0, 1.0, M0=5, fpm=fpm.copy(), engine=engine, contour=(Zne, Wne))
    want = [x for x in d if abs(x - 2.0) <= 1.0]
    assert base.info == 0 and rx.info == 0 and rx.M == base.M == len(want)
    assert np.allclose(sorted(rx.lambda_, key=ckey), sorted(want, key=ckey), atol=1e-9)
