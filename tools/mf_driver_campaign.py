"""One-off campaign: the Hermitian driver fuzz cases of tests/test_gpu_driver_fuzz.py with every shifted system solved by the
MULTIFRONTAL plan of the sparse direct solver (FH_MF=1, solver="banded"), loop for loop against the oracle's sparse LU per node.
Dense cases go in as CSR matrices with a full pattern (one dense front).  Usage: python tools/mf_driver_campaign.py [first_seed] [n_seeds] [cases_per_seed]"""
import os, sys
os.environ["FH_MF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp
import feast_oracle as fo
import feastkit_jl_amd as fk
import test_gpu_driver_fuzz as t
first = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 8
eng = fk.HipEngine(0)
bad = total = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    done = 0
    while done < cases:
        c = t._case(rng)
        if c is None:
            continue
        kind, A, B, Ad, Bd, want, Emin, Emax, M0 = c
        As = sp.csr_matrix(A) if not sp.issparse(A) else A.tocsr()
        Bs = None if B is None else (sp.csr_matrix(B) if not sp.issparse(B) else B.tocsr())
        N = Ad.shape[0]
        Q0 = fo.seeded_subspace(N, M0, seed=seed + done)
        fpm = fk.feastinit(); fpm[2] = 8; fpm[4] = 40
        got = fk.feast(As, Bs, (Emin, Emax), M0=M0, fpm=fpm, engine=eng, Q0=Q0, solver="banded")
        ref = fo.feast_hermitian(Ad, Bd, Emin, Emax, M0, ne=8, fpm4=40, Q0=Q0, real_projection=True)
        tag = f"seed={seed} case={done} kind={kind} N={N} gen={B is not None} k={len(want)} M0={M0} plan={eng.band_plan()[3]}"
        total += 1
        ok = (got.info, got.M) == (ref.info, ref.M) and eng.band_plan()[3] == 2
        if ok and ref.info == 0:
            scale = max(1.0, np.abs(want).max())
            ok = got.M == len(want) and np.abs(np.sort(got.lambda_) - want).max() <= 1e-9 * scale and abs(got.loop - ref.loop) <= 1
            if ok:
                BX = got.q if Bd is None else Bd @ got.q
                res = np.linalg.norm(Ad @ got.q - BX * got.lambda_, axis=0) / np.maximum(np.abs(got.lambda_), 1.0) / np.linalg.norm(got.q, axis=0)
                ok = res.max() <= 1e-10
        elif ok:
            ok = got.loop == ref.loop
        if not ok:
            bad += 1
            print("FAILED:", tag, "got", got.info, got.M, got.loop, "ref", ref.info, ref.M, ref.loop, flush=True)
        done += 1
    print("seed", seed, "done", flush=True)
print(f"multifrontal driver campaign: {total} cases, {bad} failures")
sys.exit(1 if bad else 0)
