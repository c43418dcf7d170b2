#!/usr/bin/env python3
"""Experiment: the contour policy (hip_backend.feast_hip_hermitian contour_policy="auto") against fixed ellipse ratios on
cfg 3 and on the non-commuting pencils of workloads.variable_coefficient_pencil.  One JSON line per run."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse.linalg as spla
import torch
import feastkit_jl_amd as fk

which = sys.argv[1:] or ["cfg3", "diag3d", "stiff3d", "diag2d"]
eng = fk.HipEngine(0)
M0 = 64


def problem(tag):
    if tag == "cfg3":
        A, B, lam = fk.workloads.laplacian_3d_pencil(50, 40, 25, 0.1)
        return A, B, 0.0, 0.1775, 44
    dims, kind = {"diag3d": ((50, 40, 25), "diag_mass"), "stiff3d": ((50, 40, 25), "stiff_mass"), "diag2d": ((250, 200), "diag_mass")}[tag]
    A, B = fk.workloads.variable_coefficient_pencil(dims, kind)
    w = np.sort(spla.eigsh(A, k=50, M=B, sigma=0.0, which="LM", return_eigenvectors=False))
    return A, B, 0.0, 0.5 * (w[43] + w[44]), 44


for tag in which:
    A, B, Emin, Emax, want = problem(tag)
    eng.set_problem(A, B)
    Q0 = eng.upload(fk.seeded_subspace(A.shape[0], M0))
    only = os.environ.get("EXP_RUNS")
    runs = [("circle cap100", dict(f18=100, cap=100)), ("a=4000 cap50", dict(f18=4000, cap=50)), ("a=2400 cap50", dict(f18=2400, cap=50)),
            ("auto cap100", dict(policy="auto", cap=100)), ("auto cap50", dict(policy="auto", cap=50))]
    for name, kw in runs:
        if only and not any(name.startswith(o) for o in only.split(",")):
            continue
        best, out = 1e9, None
        for rep in range(2):
            fpm = fk.feastinit()
            fpm[2], fpm[4] = 16, 40
            if "f18" in kw:
                fpm[18] = kw["f18"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fk.feast_hip_hermitian(eng, A, B, Emin, Emax, M0, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2,
                                         solver_maxiter=kw["cap"], preloaded=True, Q0=Q0, real_projection=True,
                                         contour_policy=kw.get("policy"))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        hres = float("nan")
        if out.M:
            Bq = out.q if B is None else B @ out.q
            hres = float((np.linalg.norm(A @ out.q - Bq * out.lambda_, axis=0) / np.maximum(np.abs(out.lambda_), 1.0)).max())
        print(json.dumps({"problem": tag, "run": name, "ms": round(1e3 * best, 1), "eig_per_s": round(out.M / best, 1), "M": out.M, "want": want,
                          "info": out.info, "loops": out.loop, "its": out.stats["krylov_iterations"], "epsout": out.epsout, "host_res": hres,
                          "eps": ["%.1e" % l["epsout"] for l in out.stats["loops"]],
                          "policy": out.stats.get("contour_policy")}), flush=True)
