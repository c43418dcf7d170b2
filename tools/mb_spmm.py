"""Time the CSR SpMM alone (feasthip_matmul on cfg 3, one node, 64 columns).  Usage: python tools/mb_spmm.py [reps] [columns]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import feastkit_jl_amd as fk
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
A, B, lam = fk.workloads.laplacian_3d_pencil(50, 40, 25)
eng = fk.HipEngine(0)
eng.set_problem(A, B)
X = eng.upload(fk.seeded_subspace(50000, m))
eng.matmul(0, X, m)
eng.profile_reset(); eng.profile_enable(True)
for _ in range(reps):
    eng.matmul(0, X, m)
ms, n = eng.profile_get("spmm")
print(f"k_spmm one node, {m} columns: {ms / max(n, 1) * 1e3:.1f} us per launch ({n} launches)")
