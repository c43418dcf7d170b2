"""The opt-in LDS-window SpMM path (ingest renumbering into 128-row blocks + k_spmm_lds, fh_sparse.hip / fh_api.hip;
FH_LDS_SPMM=1).  The kernel is off by default -- it measured slower than the gather kernels on cfg 3
-- but it is the north_star's "CSR SpMV staging rows through LDS" and stays correct; the renumbering itself is ON by
default for wide patterns since round 3 (it feeds the row-per-wave kernel).  A worker process with FH_REORDER=2 pushes
matrix products, Krylov solves, a contour sweep and a full FEAST solve through it and compares with numpy / the
closed form.  The renumbering must be invisible: every result is for the matrix as the caller defined it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path[:0] = [r"{root}", r"{root}/oracle", r"{root}/tests"]
import numpy as np, scipy.sparse as sp
import feast_oracle as fo, feastkit_jl_amd as fk
eng = fk.HipEngine(0)
rng = np.random.default_rng(5)
def block(N, m, seed):
    r = np.random.default_rng(seed)
    return np.asfortranarray(r.standard_normal((N, m)) + 1j * r.standard_normal((N, m)))
# 1. products: real symmetric pair with far couplings (long rows, outside rows beyond the kept list), complex pair, B = I
for N, m, cplx, bid in ((700, 7, False, False), (3000, 64, False, False), (2500, 33, True, False), (1500, 16, False, True)):
    A = sp.random(N, N, density=min(1.0, 9.0 / N), random_state=N, format="csr")
    A = A + A.T + sp.diags(np.arange(1, N + 1, dtype=float))
    dense_row = np.zeros(N); dense_row[::7] = 0.01                      # one long row/column: > 16 nonzeros, > 160 outside rows
    A = sp.lil_matrix(A); A[5, :] = A[5, :] + dense_row; A[:, 5] = A[:, 5] + dense_row[:, None]; A = sp.csr_matrix(A)
    if cplx:
        S = sp.random(N, N, density=min(1.0, 3.0 / N), random_state=N + 1, format="csr")
        A = sp.csr_matrix(A + 1j * (S - S.T))
    B = None if bid else sp.csr_matrix(sp.identity(N) + 0.1 * abs(A))
    eng.set_problem(A, B)
    X = block(N, m, N + 2)
    for which, M in ((0, A), (1, sp.identity(N) if B is None else B)):
        Y = eng.download(eng.matmul(which, eng.upload(X), m), m)
        ref = M @ X
        assert np.abs(Y - ref).max() <= 1e-12 * np.abs(ref).max(), (N, m, which)
# 2. Krylov solves + contour sweep on a real-symmetric pencil against dense LAPACK
N, m = 1200, 20
A, B, lam = fo.cfg3_problem(12, 10, 10)
Ad, Bd = A.toarray(), B.toarray()
eng.set_problem(A, B)
X = block(N, m, 3)
for solver in ("cocg", "bicgstab", "gmres"):
    eng.set_solver(solver, rtol=1e-12, atol=0.0, maxit=4000, restart=40)
    z = 0.3 + 0.2j
    dY, rc = eng.shifted_solve(z, eng.upload(X), m)
    assert rc == 0
    ref = np.linalg.solve(z * Bd - Ad, X)
    assert np.abs(eng.download(dY, m) - ref).max() <= 1e-8 * np.abs(ref).max(), solver
fpm = fk.feastdefault(fk.feastinit())
Z, W = fk.feast_contour(0.0, 0.6, fpm)
eng.set_contour(Z, W, 2.0); eng.set_real_projection(True); eng.set_node_range(0, len(Z))
eng.set_solver("cocg", rtol=1e-12, atol=0.0, maxit=4000)
Q = np.asfortranarray(np.real(X).astype(np.complex128))
dP, status, st = eng.contour_apply(eng.upload(Q), m)
ref = sum(2 * w * np.linalg.solve(z * Bd - Ad, Bd @ Q) for z, w in zip(Z, W)).real
assert int(status.max()) == 0 and np.abs(eng.download(dP, m).real - ref).max() <= 1e-9 * np.abs(ref).max()
# 3. the whole solve, bench settings, eigenvalues in closed form
inside = lam[(lam >= 0.0) & (lam <= 0.6)]
fpm = fk.feastinit(); fpm[2], fpm[4], fpm[18] = 16, 40, 1000
r = fk.feast_hip_hermitian(eng, A, B, 0.0, 0.6, len(inside) + 12, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2,
                           solver_maxiter=80, real_projection=True)
assert r.info == 0 and r.M == len(inside) and np.abs(np.sort(r.lambda_) - inside).max() <= 1e-10
res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0)
assert res.max() <= 1e-10
print("lds path ok")
'''


def test_lds_window_spmm_path(engine, tmp_path):
    script = tmp_path / "lds_worker.py"
    script.write_text(WORKER.format(root=ROOT))
    for lds in ("1", "0"):        # the LDS-window kernel, then the gather kernels on the same renumbered matrices
        env = dict(os.environ, FH_REORDER="2", FH_LDS_SPMM=lds, FH_DEBUG_TIMING="1")
        p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
        out = p.stdout.decode()
        assert p.returncode == 0 and "lds path ok" in out, out
        assert "renumbered into" in out          # the path under test really ran


@pytest.mark.parametrize("env", [{"FH_REORDER": "0"}, {"FH_COCG_FUSED": "0"}, {"FH_SPMM_ROW": "0"},
                                 {"FH_REORDER": "0", "FH_COCG_FUSED": "0", "FH_SPMM_ROW": "0"}, {"FH_LU_3M": "0"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_comparison_switches_stay_correct(engine, tmp_path, env):
    """The environment switches that select the round-2 forms (caller's row order, five-launch COCG iteration,
    4-rows-per-wave SpMM, four-product LU update) exist so that every comparison quoted in DESIGN.md can be re-run on one
    build: the same worker (products, Krylov solves against dense LAPACK, a contour sweep, a full FEAST solve) must pass
    under each of them."""
    script = tmp_path / "switch_worker.py"
    script.write_text(WORKER.format(root=ROOT))
    p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, **env), timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "lds path ok" in out, out
