#!/usr/bin/env python3
"""Experiment: cfg 3 solve time against contour type (fpm[16]), aspect (fpm[18]), inner tolerance and
iteration cap.  One line per setting: ms per solve, loops, Krylov node-iterations, M, device residual."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import feastkit_jl_amd as fk

A, B, lam = fk.workloads.laplacian_3d_pencil(50, 40, 25, 0.1)
Emin, Emax, M0 = 0.0, 0.1775, 64
inside = lam[(lam >= Emin) & (lam <= Emax)]
eng = fk.HipEngine(0)
eng.set_problem(A, B)
Q0 = eng.upload(fk.seeded_subspace(A.shape[0], M0))


def run(tag, fpm16=0, fpm18=100, rtol=3e-2, cap=100, ne=16, solver="cocg", reps=2, **kw):
    out = None
    best = 1e9
    for _ in range(reps):
        fpm = fk.feastinit()
        fpm[2], fpm[4], fpm[16], fpm[18] = ne, 40, fpm16, fpm18
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fk.feast_hip_hermitian(eng, A, B, Emin, Emax, M0, fpm, solver=solver, warm_start=True, inner_rtol=rtol,
                                     solver_maxiter=cap, preloaded=True, Q0=Q0, real_projection=True, **kw)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    err = float(np.abs(np.sort(out.lambda_) - inside).max()) if out.M == len(inside) else float("nan")
    its = [l["krylov_iterations"] for l in out.stats["loops"]]
    print(json.dumps({"tag": tag, "ms": round(best * 1e3, 1), "loops": out.loop, "M": out.M, "info": out.info, "epsout": out.epsout,
                      "err": err, "its": out.stats["krylov_iterations"], "its_per_loop": its,
                      "phase": {k: round(v, 4) for k, v in out.stats["phase_seconds"].items()}}), flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
run("gauss a=1 (bench)")
if which in ("all", "contour"):
    run("trapezoid a=1", fpm16=1)
    run("zolotarev", fpm16=2)
    for a in (50, 150, 200, 300):
        run("gauss a=%.2f" % (a / 100), fpm18=a)
    for a in (50, 150, 200, 300):
        run("trapezoid a=%.2f" % (a / 100), fpm16=1, fpm18=a)
if which == "tall":
    for a in (800, 1200, 1600, 2400, 3200, 4000, 5000):
        run("gauss a=%.0f" % (a / 100), fpm18=a)
    for a, rtol, cap in ((2400, 3e-2, 50), (4000, 3e-2, 50), (4000, 1e-2, 100), (4000, 1e-1, 100), (3200, 3e-2, 60), (4000, 3e-2, 200)):
        run("gauss a=%.0f rtol=%g cap=%d" % (a / 100, rtol, cap), fpm18=a, rtol=rtol, cap=cap)
    for a in (300, 400):
        run("trapezoid a=%.0f" % (a / 100), fpm16=1, fpm18=a)
if which in ("all", "tol"):
    for f16 in (0, 1):
        for rtol, cap in ((1e-1, 100), (1e-1, 50), (3e-2, 50), (3e-2, 200), (1e-2, 100), (1e-2, 200), (1e-3, 300), (1e-4, 500)):
            run("fpm16=%d rtol=%g cap=%d" % (f16, rtol, cap), fpm16=f16, rtol=rtol, cap=cap)
