import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import feastkit_jl_amd as fk
from feastkit_jl_amd import hip_backend as hb
import bench
torch.cuda.set_device(0)
eng = fk.HipEngine(0)
A, B, lam = bench.build_problem()
eng.set_problem(A, B)
fpm = fk.feastinit(); fpm[2]=16; fpm[16]=0; fpm[18]=4000
Zne, Wne = fk.contour.feast_contour(bench.EMIN, bench.EMAX, fpm) if hasattr(fk,'contour') else (None,None)
N, M0 = A.shape[0], 64
rng = np.random.default_rng(0)
Q = rng.standard_normal((N, M0)) + 0j
dP0 = eng.upload(Q)
def T(f, n=40):
    f(); eng.synchronize()
    t=time.perf_counter()
    for _ in range(n): f()
    eng.synchronize()
    return (time.perf_counter()-t)/n*1e6
print("sync_stream only      %.1f us" % T(lambda: eng._sync_stream()))
print("torch empty 51MB      %.1f us" % T(lambda: eng.empty(M0)))
dP = dP0.clone()
def ortho():
    global dP
    dP.copy_(dP0)
    return eng.orthonormalize(dP, M0, 1.5e-8)
print("copy_ 51MB            %.1f us" % T(lambda: dP.copy_(dP0)))
print("ortho (+copy)         %.1f us" % T(ortho))
r = ortho()
print("rank", r)
print("project               %.1f us" % T(lambda: eng.project(dP, r, bilinear=False, hermitize=True)))
Sq, Aq = eng.project(dP, r, bilinear=False, hermitize=True)
print("reduced eig (host)    %.1f us" % T(lambda: hb._reduced_hermitian_eig(Sq, Aq)))
lam_red, v_red = hb._reduced_hermitian_eig(Sq, Aq)
perm, M = hb._reorder_by_interval(lam_red, bench.EMIN, bench.EMAX, r)
V = np.asfortranarray(v_red[:, perm]); ls = lam_red[perm]
print("reorder               %.1f us" % T(lambda: (hb._reorder_by_interval(lam_red, bench.EMIN, bench.EMAX, r), np.asfortranarray(v_red[:, perm]))))
M = max(M, 44)
print("ritz_residual         %.1f us" % T(lambda: eng.ritz_residual(dP, r, V, ls, M, normalize=True, use_B=True)))
for cls in ("ortho","gram","project","ritz","spmm","small_matmul","residual"):
    pass
eng.profile_reset(); eng.profile_set_period(1); eng.profile_enable(True)
for _ in range(10):
    ortho(); eng.project(dP, r); eng.ritz_residual(dP, r, V, ls, M, normalize=True, use_B=True)
eng.profile_enable(False)
import ctypes
for cls in ("gram","ortho","small_matmul","spmm","project","ritz","resid","normalize","to_panel","from_panel","blockops","dense_op","axpy","scale"):
    try:
        ms, n = eng.profile_get(cls)
        if n: print(f"  class {cls:12s} launches/10 {n/10:5.1f}  ms per RR {ms/10:7.3f}")
    except Exception as e: pass
# ---- the same step through the resident entry points (round 4) ----
print("--- resident ---")
eng.import_resident(dP0, M0, which=1)
def reduce_res():
    eng.import_resident(dP0, M0, which=1)
    return eng.rr_reduce_resident(M0, 1.5e-8)
print("import (to_panel+sync) %.1f us" % T(lambda: eng.import_resident(dP0, M0, which=1)))
print("rr_reduce (+import)   %.1f us" % T(reduce_res))
r2, Sq2, Aq2 = reduce_res()
l2, v2 = hb._reduced_hermitian_eig(Sq2, Aq2)
p2, M2 = hb._reorder_by_interval(l2, bench.EMIN, bench.EMAX, r2)
V2 = np.asfortranarray(v2[:, p2]); ls2 = l2[p2]
print("rank", r2, "max |lambda - legacy|", float(np.abs(np.sort(l2) - np.sort(lam_red)).max()))
print("rr_ritz               %.1f us" % T(lambda: eng.rr_ritz_resident(r2, V2, ls2, max(M2, 44))))
