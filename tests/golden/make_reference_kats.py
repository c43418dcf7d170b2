"""Writes tests/golden/reference_kats.json: the known-answer fixtures the reference's own
test-suite holds for the hot path, transcribed as DATA (inputs + expected outputs; the
reference is Julia and cannot be executed in the build image, so nothing here was produced
by running it).  Expected values are either literals of the reference tests or closed forms
they compare against (LAPACK eigvals of tiny matrices == closed form here).

Sources (reference checkout):  test/runtests.jl:155-238, 415-432, 552-580, 1042-1089;
test/test_allocation_helpers.jl:85-209, 219-265, 274-328; test/test_parallel_backends.jl:10-48,
90-141.  Run:  python tests/golden/make_reference_kats.py
"""
import json
import math
import os

import numpy as np


def c(z):
    z = complex(z)
    return [z.real, z.imag]


def cm(M):
    return [[c(v) for v in row] for row in np.asarray(M)]


kats = {}

# --- helper KATs (test/test_allocation_helpers.jl) -----------------------------------------
kats["reorder_by_interval"] = {   # :85-107
    "lambda": [4.0, 2.0, 1.0, 3.0], "Emin": 1.5, "Emax": 3.5,
    "vectors": [[11, 12, 13, 14], [21, 22, 23, 24], [31, 32, 33, 34]],
    "expect_m": 2, "expect_lambda": [2.0, 3.0, 4.0, 1.0], "expect_cols_1based": [2, 4, 1, 3]}
kats["feast_sort"] = {            # :131-150
    "lambda": [4.0, 2.0, 1.0, 3.0], "res": [0.4, 0.2, 0.1, 0.3],
    "q": [[11.0, 12.0, 13.0, 14.0], [21.0, 22.0, 23.0, 24.0], [31.0, 32.0, 33.0, 34.0]],
    "expect_lambda": [1.0, 2.0, 3.0, 4.0], "expect_cols_1based": [3, 2, 4, 1], "expect_res": [0.1, 0.2, 0.3, 0.4]}
kats["feast_sort_general"] = {    # :152-181
    "lambda": [c(4.0), c(1.0 + 1.0j), c(0.5), c(2.0)], "res": [0.4, 0.2, 0.1, 0.3],
    "expect_order_1based": [3, 2, 4, 1], "expect_res": [0.1, 0.2, 0.3, 0.4]}
A = np.array([[4.0, 0.2, 0.0], [0.2, 5.0, 0.3], [0.0, 0.3, 6.0]])
B = np.diag([1.0, 1.2, 1.5])
q = np.array([[0.8, 0.1], [0.3, 0.7], [0.5, 0.6]])
lam = [4.2, 5.8]
kats["feast_residual"] = {        # :183-209, expected = closed form the test itself evaluates
    "A": A.tolist(), "B": B.tolist(), "q": q.tolist(), "lambda": lam,
    "expect": [float(np.linalg.norm(A @ q[:, j] - lam[j] * (B @ q[:, j])) / max(abs(lam[j]), 1.0)) for j in range(2)]}
work = np.array([[1.0 + 0.2j, 0.3 - 0.1j], [0.4 - 0.3j, 1.1 + 0.5j], [0.7 + 0.4j, 0.2 + 0.9j]])
workc = np.array([[0.8 - 0.2j, 0.6 + 0.3j], [0.1 + 0.7j, 1.2 - 0.4j], [0.5 - 0.6j, 0.4 + 0.2j]])
kats["moment_accumulation"] = {   # :219-265: Aq = Wne[1]*(work'*workc), Bq = Zne[1]*Aq
    "work": cm(work), "workc": cm(workc), "Zne1": c(0.5), "Wne1": c(0.25),
    "expect_Aq": cm(0.25 * (work.conj().T @ workc)), "expect_Bq": cm(0.5 * 0.25 * (work.conj().T @ workc))}
src = np.array([[1.0, 2.0, 0.0, 1.0e-15], [1.0j, 2.0j, 1.0, 1.0e-15j], [0, 0, 1.0j, 0], [0, 0, 0, 0]], dtype=complex)
kats["qr_compress"] = {"src": cm(src), "ncols": 4, "expect_rank": 2, "span_tol": 1e-12}   # :274-292
kats["shifted_identity"] = {"n": 120, "z": c(1.5 + 0.25j)}   # :479-500 (z*I - A, same nnz)

# --- integration KATs ------------------------------------------------------------------------
tri3 = [2 - 2 * math.cos(k * math.pi / 4) for k in (1, 2, 3)]
kats["tridiag3_real_sym"] = {     # runtests.jl:155-163
    "n": 3, "interval": [0.5, 3.5], "M0": 3, "expect_lambda": tri3, "atol": 1e-10, "expect_M": 3, "expect_info": 0}
H = np.array([[2.5, 0.2 + 0.1j, 0.0], [0.2 - 0.1j, 3.5, 0.3 - 0.2j], [0.0, 0.3 + 0.2j, 4.0]])
kats["hermitian3_dense"] = {      # :171-178
    "A": cm(H), "interval": [2.0, 5.0], "M0": 3, "expect_lambda": np.linalg.eigvalsh(H).tolist(), "atol": 1e-9}
v = np.array([0.1 + 0.2j, -0.05 + 0.1j])
S = np.diag([2.0, 3.0, 4.0]).astype(complex) + np.diag(v, 1) + np.diag(v.conj(), -1)
kats["hermitian3_sparse"] = {     # :187-194
    "A": cm(S), "interval": [1.5, 4.5], "M0": 3, "expect_lambda": np.linalg.eigvalsh(S).tolist(), "atol": 1e-9}
kats["general2"] = {              # :204-222
    "A": cm([[1, 2 + 1j], [0, 3]]), "B": cm([[1, 0], [0, 2]]), "center": c(2.0), "radius": 2.5, "M0": 2,
    "expect_standard": [1.0, 3.0], "expect_generalized": [1.0, 1.5], "atol": 1e-9}
kats["diag80_oversized"] = {      # test_allocation_helpers.jl:294-328
    "n": 80, "interval": [10.5, 12.5], "M0": 32, "fpm2": 8, "fpm3": 7, "fpm4": 4,
    "expect_lambda": [11.0, 12.0], "atol": 1e-8, "max_res": 1e-7, "expect_M": 2, "expect_info": 0}
kats["diag4_variant_b"] = {       # runtests.jl:1042-1089
    "diag": [0.5, 1.0, 1.5, 3.0], "interval": [0.4, 1.6], "fpm2": 8, "fpm4": 12, "M0": 4,
    "expect_lambda": [0.5, 1.0, 1.5], "atol": 1e-8, "expect_M": 3}
kats["tridiag10_backends"] = {    # test_parallel_backends.jl:10-48, test_backend_api.jl:9-18
    "n": 10, "interval": [0.1, 3.9], "fpm2": 8, "fpm4": 20, "M0": 10,
    "expect_lambda": [2 - 2 * math.cos(k * math.pi / 11) for k in range(2, 10)], "atol": 1e-8}
kats["hermitian_generalized_diag6"] = {   # runtests.jl:415-432
    "A_diag": [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], "B_diag": [1.0, 1.2, 1.5, 2.5, 4.0, 5.0], "interval": [0.5, 3.1], "M0": 6,
    "expect_lambda": sorted(x for x in (a / b for a, b in zip([1, 2, 3, 4, 5, 6], [1.0, 1.2, 1.5, 2.5, 4.0, 5.0])) if 0.5 <= x <= 3.1),
    "atol": 1e-8}
kats["gmres_equiv_tridiag12"] = {  # runtests.jl:552-580
    "n": 12, "interval": [0.1, 3.9], "M0": 12, "solver_tol": 1e-6, "maxiter": 400, "restart": 20, "atol": 1e-6,
    "expect_lambda": [x for x in (2 - 2 * math.cos(k * math.pi / 13) for k in range(1, 13)) if 0.1 <= x <= 3.9]}
kats["mpi_complex_hermitian_diag4"] = {   # test_parallel_backends.jl:90-123
    "diag": [0.5, 1.0, 1.5, 3.0], "interval": [0.4, 1.6], "fpm2": 8, "fpm4": 12, "expect_lambda": [0.5, 1.0, 1.5], "atol": 1e-8}
kats["mpi_complex_general_diag4"] = {     # test_parallel_backends.jl:125-141
    "diag": [c(0.5 + 0.1j), c(1.0 + 0.2j), c(2.0 - 0.1j), c(4.0)], "center": c(1.0 + 0.1j), "radius": 1.3,
    "fpm3": 11, "fpm4": 12, "fpm8": 12, "expect_lambda": [c(0.5 + 0.1j), c(1.0 + 0.2j), c(2.0 - 0.1j)], "atol": 1e-8}
kats["contour_lengths"] = {"fpm2_default": 8, "fpm8_default": 16}   # runtests.jl:48-57
kats["fpm_defaults"] = {"1": 0, "2": 8, "3": 12, "4": 20}             # runtests.jl:10-46

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
with open(out, "w") as f:
    json.dump(kats, f, indent=1)
print("wrote", out, len(kats), "fixtures")
