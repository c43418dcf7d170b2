"""Fixed-iteration contour_apply on cfg 3 for kernel timing / PMC passes.
Usage: python tools/mb_apply.py [nodes] [iterations] [reps]   (env SOLVER, PREC, M = columns)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import feastkit_jl_amd as fk
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
A, B, lam = fk.workloads.laplacian_3d_pencil(50, 40, 25)
eng = fk.HipEngine(0)
eng.set_problem(A, B)
fpm = fk.feastdefault(fk.feastinit()); fpm[2] = 16
Z, W = fk.feast_contour(0.0, 0.1775, fpm)
eng.set_contour(Z, W, 2.0)
eng.set_node_range(0, nodes)
eng.set_solver(os.environ.get("SOLVER","bicgstab"), rtol=0.0, atol=0.0, maxit=maxit, factor_precision=int(os.environ.get("PREC","64")))
M = int(os.environ.get("M", "64"))
Q = eng.upload(fk.seeded_subspace(50000, M))
eng.contour_apply(Q, M)
eng.profile_reset(); eng.profile_enable(os.environ.get('NOPROF') is None)
t0 = time.perf_counter()
for _ in range(reps):
    dP, status, st = eng.contour_apply(Q, M)
dt = time.perf_counter() - t0
eng.profile_enable(False)
its = reps * maxit * nodes
print(f"nodes={nodes} M={M} maxit={maxit}: {dt*1e3/ (reps*maxit):.3f} ms per iteration (all nodes), {dt*1e3/its:.4f} ms per node-iteration")
for cls in ("spmm", "bicg_xr", "bicg_p", "bicg_s", "cocg_xr", "cocg_p", "cocg_vec", "dot_finalize"):
    ms, n = eng.profile_get(cls)
    print(f"  {cls:14s} launches {n:5d} avg {ms/max(n,1)*1e3:8.1f} us")
