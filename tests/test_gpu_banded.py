"""GPU tests of the batched banded LU (FEASTHIP_SOLVER_BANDED, ZGBTRF/ZGBTRS semantics) and of the banded
drivers (src/banded/feast_banded.jl) that sit on it."""
import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

import feast_oracle as fo
import feastkit_jl_amd as fk
from feastkit_jl_amd import ingest, rci

pytestmark = pytest.mark.gpu


def band_matrix(n, kl, ku, seed, cplx=True, weak_diag=False):
    rng = np.random.default_rng(seed)
    diags, offs = [], []
    for d in range(-kl, ku + 1):
        v = rng.standard_normal(n - abs(d))
        if cplx:
            v = v + 1j * rng.standard_normal(n - abs(d))
        if d == 0 and not weak_diag:
            v = v + 4.0
        diags.append(v); offs.append(d)
    return sp.csr_matrix(sp.diags(diags, offs, shape=(n, n)))


@pytest.mark.parametrize("n,kl,ku,m", [(12, 1, 1, 3), (200, 2, 3, 16), (1500, 7, 4, 40), (900, 12, 12, 64), (700, 3, 1, 100)])
@pytest.mark.parametrize("weak_diag", [False, True])
def test_banded_shifted_solve_matches_numpy(engine, n, kl, ku, m, weak_diag):
    """weak_diag: no diagonal dominance, the row interchanges (and the kl rows of fill-in) are exercised."""
    A = band_matrix(n, kl, ku, 3, weak_diag=weak_diag)
    B = band_matrix(n, min(kl, 1), min(ku, 1), 4)
    engine.set_problem(A, B)
    engine.set_solver("banded")
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))
    z = 0.7 + 0.9j
    dY, rc = engine.shifted_solve(z, engine.upload(X), m)
    assert rc == 0
    Y = engine.download(dY, m)
    S = (z * B - A).toarray()
    ref = np.linalg.solve(S, X)
    assert np.abs(Y - ref).max() <= 1e-9 * np.abs(ref).max() * max(1.0, np.linalg.cond(S) * 1e-6)
    assert np.linalg.norm(S @ Y - X) <= 1e-10 * np.linalg.norm(X) * max(1.0, np.linalg.cond(S) * 1e-4)


def test_banded_contour_apply_matches_oracle_and_caches(engine):
    n = 400
    A = sp.csr_matrix(sp.diags([-np.ones(n - 2) * 0.3, -np.ones(n - 1), 2 * np.ones(n) + 0.01 * np.arange(n), -np.ones(n - 1), -np.ones(n - 2) * 0.3],
                               [-2, -1, 0, 1, 2]))
    B = sp.csr_matrix(sp.diags([0.1 * np.ones(n - 1), np.ones(n) + 0.001 * np.arange(n), 0.1 * np.ones(n - 1)], [-1, 0, 1]))
    engine.set_problem(A, B)
    fpm = fk.feastdefault(fk.feastinit()); fpm[2] = 6
    Z, W = fk.feast_contour(0.4, 1.1, fpm)
    engine.set_contour(Z, W, 2.0)
    engine.set_real_projection(False)
    engine.set_solver("banded")
    Q = fk.seeded_subspace(n, 10)
    dP, status, st = engine.contour_apply(engine.upload(Q), 10)
    assert st["factorizations"] == 6 and np.all(status[:6] == 0)
    want = sum(2 * W[e] * np.linalg.solve((Z[e] * B - A).toarray(), B @ Q) for e in range(6))
    assert np.abs(engine.download(dP, 10) - want).max() <= 1e-10 * np.abs(want).max()
    dP2, status, st2 = engine.contour_apply(engine.upload(Q), 10)
    assert st2["factorizations"] == 0                      # factors cached per node
    assert np.array_equal(engine.download(dP2, 10), engine.download(dP, 10))


def test_banded_singular_shift_reports_lapack(engine):
    A = sp.csr_matrix(sp.diags([np.array([1.0, 2.0, 3.0, 4.0])], [0]))
    engine.set_problem(A, None)
    engine.set_solver("banded")
    dY, rc = engine.shifted_solve(2.0 + 0j, engine.upload(np.ones((4, 2))), 2)
    assert rc == 8


def test_sparse_direct_feast_1d_laplacian(engine):
    """feast(A, ...; solver=:direct) on a sparse band matrix: banded LU instead of UMFPACK; closed form."""
    n = 3000
    A = sp.csr_matrix(sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]))
    ev = 2 - 2 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))
    lo, hi = 0.5 * (ev[99] + ev[100]), 0.5 * (ev[129] + ev[130])
    fpm = fk.feastinit(); fpm[2] = 8
    r = fk.feast_hip_hermitian(engine, A, None, lo, hi, 45, fpm, solver="banded", real_projection=True)
    assert r.info == 0 and r.M == 30 and np.allclose(np.sort(r.lambda_), ev[100:130], atol=1e-11)
    assert r.stats["factorizations"] == 8
    o = fo.feast_hermitian(A, None, lo, hi, 45, ne=8, real_projection=True)
    assert o.info == 0 and abs(o.loop - r.loop) <= 1


def test_banded_drivers(engine):
    """feast_sbgv / feast_hbev / feast_gbev on reference band storage."""
    # real symmetric generalized, RCI kernel (configuration the moment variant handles: M0 = inside + 1)
    n = 30
    S = np.diag(2.0 * np.ones(n)) - np.diag(np.ones(n - 1), 1) - np.diag(np.ones(n - 1), -1)
    Ab = ingest.csr_to_band_upper(sp.csr_matrix(S), 1)
    fpm = fk.feastinit(); fpm[2] = 8; fpm[3] = 11; fpm[4] = 12
    r = fk.feast_sbev(Ab, 1, 0.2, 1.3, 8, fpm, engine=engine)
    ev = np.linalg.eigvalsh(S)
    inside = ev[(ev > 0.2) & (ev < 1.3)]
    want = fo.rci_symmetric(S, None, 0.2, 1.3, 8, ne=8, fpm3=11, fpm4=12)
    assert (r.info, r.M) == (want.info, want.M) == (0, len(inside)) and np.allclose(r.lambda_, inside, atol=1e-9)
    # complex Hermitian pentadiagonal, variant A on the half contour (slow filter: allow 60 loops)
    n = 80
    rng = np.random.default_rng(9)
    o1 = 0.4 * (rng.standard_normal(n - 1) + 1j * rng.standard_normal(n - 1))
    o2 = 0.2 * (rng.standard_normal(n - 2) + 1j * rng.standard_normal(n - 2))
    H = sp.diags([o2.conj(), o1.conj(), np.linspace(1, 9, n), o1, o2], [-2, -1, 0, 1, 2]).toarray()
    Hb = ingest.csr_to_band_upper(sp.csr_matrix(H), 2)
    ev = np.linalg.eigvalsh(H)
    lo, hi = 0.5 * (ev[19] + ev[20]), 0.5 * (ev[27] + ev[28])
    fpm = fk.feastinit(); fpm[2] = 8; fpm[3] = 10; fpm[4] = 60
    r = fk.feast_hbev(Hb, 2, lo, hi, 12, fpm, engine=engine)
    assert r.info == 0 and r.M == 8 and np.allclose(r.lambda_, ev[20:28], atol=1e-8)
    # general band matrix, full contour
    G = band_matrix(120, 2, 2, 21).toarray()
    Gb = np.zeros((5, 120), dtype=complex)
    for i in range(120):
        for j in range(max(0, i - 2), min(120, i + 3)):
            Gb[2 + i - j, j] = G[i, j]
    ev = np.linalg.eigvals(G)
    c = ev[np.argsort(np.abs(ev - ev.mean()))[0]]
    dist = np.sort(np.abs(ev - c))
    rad = 0.5 * (dist[5] + dist[6])
    fpm = fk.feastinit(); fpm[8] = 16; fpm[3] = 10
    r = fk.feast_gbev(Gb, 2, c, rad, 12, fpm, engine=engine)
    inside = ev[np.abs(ev - c) <= rad]
    key = lambda x: (round(x.real, 6), round(x.imag, 6))
    assert r.info == 0 and r.M == 6 and np.allclose(sorted(r.lambda_, key=key), sorted(inside, key=key), atol=1e-8)


def test_rci_server_on_banded_solver(engine):
    n = 60
    A = sp.csr_matrix(sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]))
    srv = rci.HipRciServer(engine, A, None, solver="banded")
    solve = srv.linear_solver()
    X = np.random.default_rng(0).standard_normal((n, 4))
    Y = np.zeros((n, 4), dtype=complex)
    z = 1.3 + 0.4j
    solve(Y, z, X)
    assert np.allclose((z * sp.identity(n) - A) @ Y, X, atol=1e-12)
