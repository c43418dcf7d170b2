"""Banded LU timing: N x N band matrix (kl = ku = k), 8 contour nodes, M0 right-hand sides.
Usage: python tools/run_banded.py [N] [k] [M0]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import feastkit_jl_amd as fk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M0 = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(1)
diags = [rng.standard_normal(N - abs(d)) * (0.2 if d else 1.0) + (4.0 if d == 0 else 0.0) for d in range(-k, k + 1)]
A = sp.csr_matrix(sp.diags(diags, list(range(-k, k + 1))))
A = sp.csr_matrix(0.5 * (A + A.T))
eng = fk.HipEngine(0)
eng.set_problem(A, None)
fpm = fk.feastdefault(fk.feastinit()); fpm[2] = 8
Z, W = fk.feast_contour(3.9, 4.1, fpm)
eng.set_contour(Z, W, 2.0); eng.set_solver("banded")
Q = eng.upload(fk.seeded_subspace(N, M0))
eng.profile_reset(); eng.profile_enable(True)
for rep in range(2):
    t0 = time.perf_counter(); dP, status, st = eng.contour_apply(Q, M0); dt = time.perf_counter() - t0
    print(f"sweep {rep}: {dt*1e3:.1f} ms  factorizations {st['factorizations']}")
for cls in ("band_form", "band_lu", "band_solve"):
    ms, n = eng.profile_get(cls)
    print(f"   {cls:10s} launches {n:4d} est total {ms:9.2f} ms")
