#!/usr/bin/env python3
"""Experiment: the default call on INTERIOR intervals (guards on both sides, indefinite shifted systems) of cfg 3's pencil
and of a non-commuting pencil: circle vs the contour policy.  One JSON line per run."""
import json, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import feastkit_jl_amd as fk
warnings.simplefilter("ignore")
eng = fk.HipEngine(0)
A, B, lam = fk.workloads.laplacian_3d_pencil(50, 40, 25, 0.1)
cases = []
for centre in (0.5, 2.0):
    i0 = int(np.searchsorted(lam, centre))
    lo, hi = 0.5 * (lam[i0 - 1] + lam[i0]), 0.5 * (lam[i0 + 39] + lam[i0 + 40])
    cases.append((A, B, lo, hi, lam[i0:i0 + 40]))
for A_, B_, lo, hi, want in cases:
    for name, kw in (("policy", {}), ("circle", {"f18": 100})):
        fpm = fk.feastinit(); fpm[2] = 16; fpm[4] = 60
        if "f18" in kw:
            fpm[18] = kw["f18"]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fk.feast(A_, B_, (lo, hi), M0=64, fpm=fpm, engine=eng)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ok = r.M == len(want) and np.abs(np.sort(r.lambda_) - want).max() < 1e-9
        res = float((np.linalg.norm(A_ @ r.q - (B_ @ r.q) * r.lambda_, axis=0) / np.maximum(np.abs(r.lambda_), 1)).max()) if r.M else float("nan")
        print(json.dumps({"interval": [lo, hi], "run": name, "ms": round(1e3 * dt, 1), "info": r.info, "M": r.M, "want": len(want), "ok": bool(ok),
                          "loops": r.loop, "its": r.stats.get("krylov_iterations"), "host_res": res,
                          "policy": r.stats.get("contour_policy", {}).get("fpm18_per_loop"), "cap": r.stats.get("inner_cap")}), flush=True)
