"""Larger-than-BASELINE sanity run of the cfg-3 path: N = nx*ny*nz sparse generalized pencil, 16 nodes, M0 = 64.
Usage: python tools/run_scale.py [nx ny nz] [Emax]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import feastkit_jl_amd as fk
nx, ny, nz = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 80, 60)
A, B, lam = fk.workloads.laplacian_3d_pencil(nx, ny, nz)
N = A.shape[0]
lam = np.sort(lam)
Emax = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5 * (lam[39] + lam[40])      # 40 eigenvalues inside
inside = lam[lam <= Emax]
print(f"N={N} nnz={A.nnz} interval (0,{Emax:.5f}) holds {len(inside)} eigenvalues", flush=True)
eng = fk.HipEngine(0)
fpm = fk.feastinit(); fpm[2] = 16; fpm[4] = 60
for rep in range(2):
    t0 = time.perf_counter()
    r = fk.feast(A, B, (0.0, Emax), M0=64, fpm=fpm, engine=eng, solver="cocg", warm_start=True, inner_rtol=3e-2,
                 solver_maxiter=100, real_projection=True)
    dt = time.perf_counter() - t0
    err = np.abs(np.sort(r.lambda_) - inside).max() if r.M == len(inside) else None
    res = np.linalg.norm(A @ r.q - (B @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1.0) if r.M else np.zeros(1)
    print(f"pass {rep}: info={r.info} M={r.M} loops={r.loop} epsout={r.epsout:.2e} eigerr={err} host residual={res.max():.2e} "
          f"time={dt:.2f}s its={r.stats['krylov_iterations']} inner_cap={r.stats.get('inner_cap')}", flush=True)
