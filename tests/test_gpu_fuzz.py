"""Randomised sweep of the C-ABI primitives over awkward shapes (N = 1 ... 1023, m = 1 ... 130 incl. panel
boundaries 15/16/17, 63/64/65 and the wide path; dense and CSR; real and complex; B or identity; LU in both
precisions, banded LU, BiCGStab; rank-deficient orthonormalisation; projection; Ritz residuals), each checked
against numpy.  Deterministic seeds."""
import numpy as np
import pytest
import scipy.sparse as sp

import feastkit_jl_amd as fk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,trials", [(1, 25), (7, 25)])
def test_primitive_fuzz(engine, seed, trials):
    eng = engine
    rng = np.random.default_rng(seed)
    failures = []

    def check(name, ok, info=""):
        if not ok:
            failures.append(f"{name} {info}")

    for trial in range(trials):
        N = int(rng.choice([1, 2, 3, 5, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 129, 255, 257, 500, 1023]))
        m = int(min(N, rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 70, 130])))
        cplx = bool(rng.integers(2)); dense = bool(rng.integers(2)); bid = bool(rng.integers(2))
        if dense:
            A = rng.standard_normal((N, N)); A = A + A.T
            if cplx: S = rng.standard_normal((N, N)); A = A + 1j * (S - S.T)
            B = None if bid else (lambda M: M @ M.T / N + np.eye(N))(rng.standard_normal((N, N)))
        else:
            dens = min(1.0, 5.0 / N)
            A = sp.random(N, N, density=dens, random_state=int(rng.integers(1 << 30)), format="csr"); A = A + A.T + sp.diags(np.arange(1.0, N + 1))
            if cplx: S = sp.random(N, N, density=dens, random_state=int(rng.integers(1 << 30)), format="csr"); A = A + 1j * (S - S.T)
            A = sp.csr_matrix(A)
            B = None if bid else sp.csr_matrix(sp.diags(2.0 + rng.random(N)) + (lambda R: R + R.T)(sp.random(N, N, density=dens / 2, random_state=int(rng.integers(1 << 30)))))
        tag = f"N={N} m={m} cplx={cplx} dense={dense} bid={bid}"
        try:
            eng.set_problem(A, B)
            Ad = A if dense else A.toarray(); Bd = np.eye(N) if B is None else (B if dense else B.toarray())
            X = rng.standard_normal((N, m)) + 1j * rng.standard_normal((N, m))
            dX = eng.upload(X)
            Y = eng.download(eng.matmul(0, dX, m), m); check("matmulA " + tag, np.abs(Y - Ad @ X).max() <= 1e-10 * (np.abs(Ad @ X).max() + 1e-300))
            Y = eng.download(eng.matmul(1, dX, m), m); check("matmulB " + tag, np.abs(Y - Bd @ X).max() <= 1e-10 * (np.abs(Bd @ X).max() + 1e-300))
            z = 0.37 + 0.81j
            Sm = z * Bd - Ad
            ref = np.linalg.solve(Sm, X)
            solvers = ["direct"] if dense else ["banded" if N <= 300 else "bicgstab"]
            for sv in solvers:
                if sv == "bicgstab": eng.set_solver(sv, rtol=1e-13, atol=0.0, maxit=20000)
                else: eng.set_solver(sv)
                dY, rc = eng.shifted_solve(z, dX, m)
                Y = eng.download(dY, m)
                rel = np.linalg.norm(Sm @ Y - X) / np.linalg.norm(X)
                check(f"solve[{sv}] " + tag, rc == 0 and rel <= 1e-9 * max(1.0, np.linalg.cond(Sm) * 1e-6), f"rc={rc} rel={rel:.2e}")
                if dense and N >= 16:
                    eng.set_solver("direct", factor_precision=32)
                    dY, rc = eng.shifted_solve(z, dX, m)
                    Y = eng.download(dY, m)
                    rel = np.linalg.norm(Sm @ Y - X) / np.linalg.norm(X)
                    check("solve[lu32] " + tag, rc == 0 and rel <= 1e-9 * max(1.0, np.linalg.cond(Sm) * 1e-4), f"rc={rc} rel={rel:.2e}")
                    eng.set_solver("direct")
            # orthonormalize + project + ritz
            r_true = int(rng.integers(1, m + 1))
            src = (rng.standard_normal((N, r_true)) + 1j * rng.standard_normal((N, r_true))) @ (rng.standard_normal((r_true, m)) + 1j * rng.standard_normal((r_true, m)))
            dQ = eng.upload(np.asfortranarray(src))
            rank = eng.orthonormalize(dQ, m, 1.5e-8)
            want_rank = min(r_true, N)
            check("ortho rank " + tag, rank == want_rank, f"rank={rank} want={want_rank}")
            if rank > 0:
                Q = eng.download(dQ)[:, :rank]
                check("ortho orth " + tag, np.abs(Q.conj().T @ Q - np.eye(rank)).max() < 1e-11)
                check("ortho span " + tag, np.linalg.norm(src - Q @ (Q.conj().T @ src)) <= 1e-9 * np.linalg.norm(src))
                Aq, Bq = eng.project(dQ, rank, bilinear=False, hermitize=False)
                check("project " + tag, np.abs(Aq - Q.conj().T @ (Ad @ Q)).max() <= 1e-10 * (np.abs(Ad).max() + 1) and np.abs(Bq - Q.conj().T @ (Bd @ Q)).max() <= 1e-10 * (np.abs(Bd).max() + 1))
                V = rng.standard_normal((rank, rank)) + 1j * rng.standard_normal((rank, rank))
                lam = rng.standard_normal(rank) + 0j
                M = int(rng.integers(0, rank + 1))
                dXr, res = eng.ritz_residual(dQ, rank, np.asfortranarray(V), lam, M, normalize=True, use_B=True)
                Xr = Q @ V
                if M: Xr[:, :M] /= np.linalg.norm(Xr[:, :M], axis=0)
                check("ritz X " + tag, np.abs(eng.download(dXr)[:, :rank] - Xr).max() <= 1e-10 * np.abs(Xr).max())
                if M:
                    rr = np.linalg.norm(Ad @ Xr[:, :M] - (Bd @ Xr[:, :M]) * lam[:M].real, axis=0) / np.maximum(np.abs(lam[:M]), 1.0)
                    check("ritz res " + tag, np.abs(res - rr).max() <= 1e-9 * (rr.max() + 1e-300), f"{np.abs(res-rr).max():.2e}")
        except Exception as ex:
            failures.append(f"exception {tag}: {ex!r}"[:300])

    eng.set_solver("direct")
    assert not failures, "\n".join(failures)
