import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import feastkit_jl_amd as fk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(20260515)
rad = 33.05 * np.sqrt(N / 8192.0) * np.sqrt(rng.random(N))
delta = rad * np.exp(2j * np.pi * rng.random(N))
U = np.triu(rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N)), 1) / np.sqrt(N)
T = np.diag(delta) + 0.05 * U
del U
def refl(M, v):
    Mv = M @ v; M -= 2 * np.outer(Mv, v.conj()); vM = v.conj() @ M; M -= 2 * np.outer(v, vM); return M
for _ in range(2):
    v = rng.standard_normal(N) + 1j * rng.standard_normal(N); v /= np.linalg.norm(v)
    T = refl(T, v)
A = T
inside = delta[np.abs(delta) <= 2.0]
print("N", N, "inside", len(inside), flush=True)
eng = fk.HipEngine(0)
fpm = fk.feastinit(); fpm[8] = ne; fpm[4] = 20
eng.profile_reset(); eng.profile_enable(True)
t0 = time.perf_counter()
r = fk.feast_hip_general(eng, A, None, 0.0, 2.0, 48, fpm)
dt = time.perf_counter() - t0
key = lambda x: (round(x.real, 7), round(x.imag, 7))
err = np.abs(np.array(sorted(r.lambda_, key=key)) - np.array(sorted(inside, key=key))).max() if r.M == len(inside) else None
print(f"info={r.info} M={r.M} loops={r.loop} epsout={r.epsout:.2e} eigerr={err} time={dt:.2f}s solve={r.stats['solve_seconds']:.2f}s fact={r.stats['factorizations']}")
for cls in ("lu_form", "lu_panel", "lu_laswp", "lu_trsm", "lu_gemm_in", "lu_gemm", "lu_invert", "lu_solve", "dense_op", "gram"):
    ms, n = eng.profile_get(cls)
    if n: print(f"   {cls:10s} launches {n:5d} est total {ms:9.2f} ms")
