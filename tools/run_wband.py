"""Sparse direct path (multifrontal plan, or reverse Cuthill-McKee + blocked band LU with FH_MF=0) on cfg 3's pencil: the plan, the
factorisation and substitution times per sweep, the whole solve on the headline interval and on an interval deep inside the
spectrum (where the Krylov sweeps cannot work).  Usage: python tools/run_wband.py [--nodes 16] [--interior]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import feastkit_jl_amd as fk  # noqa: E402
from feastkit_jl_amd import workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=16)
    ap.add_argument("--interior", action="store_true")
    ap.add_argument("--precision", type=int, default=64, help="64: complex128 band factors; 32: complex64 factors + fp64 refinement")
    ap.add_argument("--dims", type=int, nargs=3, default=(50, 40, 25))
    a = ap.parse_args()
    import torch
    A, B, lam = workloads.laplacian_3d_pencil(*a.dims)
    n = A.shape[0]
    eng = fk.HipEngine(0)
    t0 = time.perf_counter()
    eng.set_problem(A, B)
    t_ingest = time.perf_counter() - t0
    t0 = time.perf_counter()
    kl, ku, nbytes, blocked = eng.band_plan()
    t_plan = time.perf_counter() - t0
    out = {"N": n, "kl": kl, "ku": ku, "GB_per_node": nbytes / 1e9, "blocked": blocked, "precision": a.precision, "ingest_s": t_ingest, "plan_s": t_plan,
           "lu_flop_per_node": eng.direct_plan_flops(), "band_flop_per_node": 8.0 * n * kl * (kl + ku)}
    fpm = fk.feastdefault(fk.feastinit()); fpm[2] = a.nodes
    Z, W = fk.feast_contour(0.0, 0.1775, fpm)
    eng.set_contour(Z, W, 2.0)
    eng.set_real_projection(True)
    eng.set_solver("banded", rtol=1e-12, factor_precision=a.precision)
    eng.profile_enable(True)
    Q = fk.seeded_subspace(n, 64)
    dQ = eng.upload(Q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dP, status, st = eng.contour_apply(dQ, 64)
    torch.cuda.synchronize()
    out["first_sweep_s"] = time.perf_counter() - t0
    out["factorizations"] = st["factorizations"]
    t0 = time.perf_counter()
    dP, status, st = eng.contour_apply(dQ, 64)
    torch.cuda.synchronize()
    out["cached_sweep_s"] = time.perf_counter() - t0
    eng.synchronize()
    out["profile_ms"] = {c: eng.profile_get(c)[0] for c in ("wband_form", "wband_lu", "wband_solve", "mf_assemble", "mf_lu", "mf_store", "mf_solve")}
    lu_cls = "mf_lu" if blocked == 2 else "wband_lu"
    out["lu_tflops"] = eng.profile_get_work(lu_cls) / max(out["profile_ms"][lu_cls], 1e-9) / 1e9
    free, total = torch.cuda.mem_get_info()
    out["device_GB_used"] = (total - free) / 1e9
    eng.profile_enable(False)
    eng.close()
    # whole solves through the API (fresh engine inside)
    cases = [("headline", 0.0, 0.1775, 44)]
    if a.interior:
        mid = 2.0
        order = np.argsort(np.abs(lam - mid))
        r = 0.5 * (abs(lam[order[39]] - mid) + abs(lam[order[40]] - mid))
        cases.append(("interior", mid - r, mid + r, 40))
    for name, lo, hi, want in cases:
        f = fk.feastinit(); f[2] = a.nodes
        for rep in range(2):
            t0 = time.perf_counter()
            res = fk.feast(A, B, (lo, hi), M0=64, fpm=f, solver="banded", inner_precision=a.precision)
            dt = time.perf_counter() - t0
            out[f"{name}_call{rep}"] = {"s": dt, "M": int(res.M), "info": int(res.info), "loops": int(res.loop), "epsout": float(res.epsout),
                                        "eigenpairs_per_s": res.M / dt, "want": want}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
