"""cfg 2 at full size on one GPU: dense symmetric N=4096 (default), 8 Gauss nodes, M0=32, direct solves.
Usage: python tools/run_cfg2.py [N]"""
import sys, os, time
os.environ.setdefault("FH_PROF_PERIOD", "1")   # time every launch: the classes of the dense path have few launches
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import feastkit_jl_amd as fk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = np.asfortranarray(fk.workloads.reflected_diagonal(0.01 * np.arange(N)))   # column-major like a Julia Matrix
eng = fk.HipEngine(0)
upload = [0.0]
_set = eng.set_problem
def timed_set(*a, **k):
    t = time.perf_counter(); out = _set(*a, **k); eng.synchronize()
    upload[0] += time.perf_counter() - t; return out
eng.set_problem = timed_set
for rep in range(2):
    fpm = fk.feastinit(); fpm[2] = 8
    eng.profile_reset(); eng.profile_enable(True)
    upload[0] = 0.0
    t0 = time.perf_counter()
    lo = 0.01 * (N // 4) - 0.005
    r = fk.feast_hip_hermitian(eng, A, None, lo, lo + 0.2, 32, fpm, solver="direct")
    dt = time.perf_counter() - t0
    eng.profile_enable(False)
    want = 0.01 * np.arange(N // 4, N // 4 + 20)
    err = np.abs(np.sort(r.lambda_) - want).max() if r.M == 20 else None
    print(f"N={N}: info={r.info} M={r.M} loops={r.loop} epsout={r.epsout:.2e} eigerr={err} time={dt:.3f}s (upload {upload[0]:.3f}s, contour sweeps {r.stats['solve_seconds']:.3f}s) phases={ {k: round(v,3) for k,v in r.stats['phase_seconds'].items()} }")
    for cls in ("lu_form", "lu_panel", "lu_laswp", "lu_trsm", "lu_gemm_in", "lu_gemm", "lu_invert", "lu_solve", "dense_op", "ortho", "gram"):
        ms, n = eng.profile_get(cls)
        if n: print(f"   {cls:10s} launches {n:5d} est total {ms:9.2f} ms")
