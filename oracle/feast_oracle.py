"""CPU oracle for the FEAST contour-integration hot path.

TEST INFRASTRUCTURE ONLY.  This file is a numpy/scipy restatement of the
reference algorithm (subhk/FeastKit.jl v1.0.11, pure Julia).  It is the checker
for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``feastkit.jl_amd``) never imports anything from ``oracle/``.

Pinning status: the reference is Julia and no ``julia`` binary exists in the
build image, so the reference itself cannot be run.  The
oracle is pinned by every known-answer fixture the reference's own test-suite
holds for this path (``tests/golden/reference_kats.json``, transcribed from
``test/runtests.jl`` and ``test/test_allocation_helpers.jl``) and by closed-form
spectra; see ``tests/test_oracle_golden.py``.

Third-party arithmetic the reference delegates to and that is restated here:
  * FastGaussQuadrature.jl 1.x ``gausslegendre(n)``  -> numpy ``leggauss`` (the
    Gauss-Legendre rule is unique; ascending node order in both).
  * LinearAlgebra ``lu/ldiv!`` (ZGETRF/ZGETRS)        -> scipy ``lu_factor/lu_solve``.
  * LinearAlgebra ``qr(A, ColumnNorm())`` (ZGEQP3)     -> scipy ``qr(pivoting=True)``.
  * LinearAlgebra ``eigen(Hermitian, Hermitian)`` (ZHEGV), ``eigen(A, B)`` (ZGGEV)
                                                      -> scipy ``eigh`` / ``eig``.
  * SparseArrays ``lu`` (UMFPACK)                     -> scipy ``splu`` (SuperLU).
  * Krylov.jl 0.10.1 ``gmres(restart=true, memory=m, rtol, atol, itmax)``
                                                      -> ``gmres_restarted`` below
    (restarted GMRES with modified Gram-Schmidt + Givens, zero initial guess,
    stop when ||r|| <= atol + rtol*||r0||).

All reference citations are ``path:line`` relative to the reference checkout.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# ---------------------------------------------------------------------------
# error / job codes -- src/core/feast_types.jl:227-268
# ---------------------------------------------------------------------------
FEAST_SUCCESS = 0
FEAST_ERROR_N = 1
FEAST_ERROR_M0 = 2
FEAST_ERROR_EMIN_EMAX = 3
FEAST_ERROR_EMID_R = 4
FEAST_ERROR_NO_CONVERGENCE = 5
FEAST_ERROR_MEMORY = 6
FEAST_ERROR_INTERNAL = 7
FEAST_ERROR_LAPACK = 8
FEAST_ERROR_FPM = 9


@dataclass
class FeastResult:
    """src/core/feast_types.jl:85-108 (FeastResult / FeastGeneralResult)."""
    lam: np.ndarray
    q: np.ndarray
    M: int
    res: np.ndarray
    info: int
    epsout: float
    loop: int
    stats: dict = field(default_factory=dict)


# ---------------------------------------------------------------------------
# a1. contours -- src/core/feast_tools.jl:212-371
# ---------------------------------------------------------------------------
def feast_contour(Emin, Emax, ne=8, fpm16=0, fpm18=100):
    """Half contour for Hermitian problems (src/core/feast_tools.jl:212-284).

    Returns (Zne, Wne) complex128 arrays of length ``ne``.
    fpm16: 0 Gauss-Legendre, 1 trapezoid, 2 Zolotarev (constants of the reference's table
    src/core/feast_tools.jl:50-180, read from the data file the package ships; :263-266).
    """
    r = (Emax - Emin) / 2.0
    Emid = Emin + r
    aspect = fpm18 * 0.01
    ba, ab = -math.pi / 2, math.pi / 2
    Zne = np.empty(ne, dtype=np.complex128)
    Wne = np.empty(ne, dtype=np.complex128)
    if fpm16 == 0:
        x, w = np.polynomial.legendre.leggauss(ne)
        for e in range(ne):
            theta = ba * x[e] + ab
            Zne[e] = Emid + r * math.cos(theta) + 1j * r * aspect * math.sin(theta)
            jac = r * 1j * math.sin(theta) + r * aspect * math.cos(theta)
            Wne[e] = 0.25 * w[e] * jac
    elif fpm16 == 1:
        for e in range(ne):
            theta = math.pi - (math.pi / ne) / 2 - (math.pi / ne) * e
            Zne[e] = Emid + r * math.cos(theta) + 1j * r * aspect * math.sin(theta)
            jac = r * 1j * math.sin(theta) + r * aspect * math.cos(theta)
            Wne[e] = (1.0 / (2 * ne)) * jac
    elif fpm16 == 2:
        import json
        import os
        # the oracle's OWN copy of the reference's constants (src/core/feast_tools.jl:50-180), extracted by
        # tests/golden/make_zolotarev_tables.py with a parser of its own -- never the file the product ships
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "zolotarev_tables.json")
        tab = json.load(open(path))[str(ne)]
        for e in range(ne):
            xr, xi, wr, wi = tab["nodes"][e]
            Zne[e] = complex(xr, xi) * r + Emid
            Wne[e] = complex(wr, wi) * r
    else:
        raise ValueError("fpm16 must be 0 (Gauss), 1 (trapezoid) or 2 (Zolotarev)")
    return Zne, Wne


def feast_gcontour(Emid, r, ne=16, fpm16=0, fpm18=100, fpm19=0):
    """Full contour for general problems (src/core/feast_tools.jl:286-371)."""
    Emid = complex(Emid)
    aspect = fpm18 * 0.01
    rot = (fpm19 / 180.0) * math.pi
    nr = r * (math.cos(rot) + 1j * math.sin(rot))
    ba, ab = -math.pi / 2, math.pi / 2
    Zne = np.empty(ne, dtype=np.complex128)
    Wne = np.empty(ne, dtype=np.complex128)
    if fpm16 == 0:
        nu = ne // 2
        xu, wu = np.polynomial.legendre.leggauss(nu) if nu > 0 else (np.zeros(0), np.zeros(0))
        xl, wl = np.polynomial.legendre.leggauss(ne - nu)
        for e in range(nu):
            theta = ba * xu[e] + ab
            Zne[e] = Emid + nr * math.cos(theta) + nr * 1j * aspect * math.sin(theta)
            jac = nr * 1j * math.sin(theta) + nr * aspect * math.cos(theta)
            Wne[e] = 0.25 * wu[e] * jac
        for e in range(nu, ne):
            i = e - nu
            theta = -ba * xl[i] - ab
            Zne[e] = Emid + nr * math.cos(theta) + nr * 1j * aspect * math.sin(theta)
            jac = nr * 1j * math.sin(theta) + nr * aspect * math.cos(theta)
            Wne[e] = 0.25 * wl[i] * jac
    else:
        for e in range(ne):
            theta = math.pi - (2 * math.pi / ne) / 2 - (2 * math.pi / ne) * e
            Zne[e] = Emid + nr * math.cos(theta) + nr * 1j * aspect * math.sin(theta)
            jac = nr * 1j * math.sin(theta) + nr * aspect * math.cos(theta)
            Wne[e] = (1.0 / ne) * jac
    return Zne, Wne


def feast_tolerance(fpm3=12):
    """src/core/feast_parameters.jl:391-396 (Float64 branch)."""
    if fpm3 < 0 or fpm3 > 16:
        return 1e-12
    return 10.0 ** (-fpm3)


# ---------------------------------------------------------------------------
# a2. seeded initial subspace -- src/core/feast_tools.jl:6-43
# (the Julia MersenneTwister stream is not reproducible outside Julia; the
#  structure -- real Gaussian columns, unit 2-norm -- is.)
# ---------------------------------------------------------------------------
def seeded_subspace(N, M0, seed=20260515, complex_values=False):
    rng = np.random.default_rng([seed, N, M0, int(complex_values)])
    Q = rng.standard_normal((N, M0))
    if complex_values:
        Q = Q + 1j * rng.standard_normal((N, M0))
    Q = Q.astype(np.complex128)
    nrm = np.linalg.norm(Q, axis=0)
    nrm[nrm == 0] = 1.0
    return np.asfortranarray(Q / nrm)


# ---------------------------------------------------------------------------
# helpers on the path -- src/core/feast_aux.jl
# ---------------------------------------------------------------------------
def hermitian_part(S):
    """src/core/feast_aux.jl:84-92."""
    return 0.5 * (S + S.conj().T)


def dense_shifted_identity_minus(z, A):
    """src/core/feast_aux.jl:59-74: z*I - A."""
    out = -np.asarray(A, dtype=np.complex128)
    out[np.diag_indices_from(out)] += z
    return out


def qr_compress(src, ncols, rank_tol=math.sqrt(np.finfo(np.float64).eps)):
    """Pivoted-QR rank compression (src/core/feast_aux.jl:101-131).

    Returns (basis[:, :rank], rank).
    """
    if ncols == 0:
        return np.zeros((src.shape[0], 0), dtype=src.dtype), 0
    blk = src[:, :ncols]
    Q, R, _ = sla.qr(blk, mode="economic", pivoting=True)
    rdiag = np.diag(R)
    if rdiag.size == 0:
        return Q[:, :0], 0
    scale = abs(rdiag[0])
    if scale == 0:
        return Q[:, :0], 0
    thr = max(rank_tol, np.finfo(np.float64).eps * max(blk.shape)) * scale
    rank = 0
    for v in rdiag:
        if not abs(v) > thr:
            break
        rank += 1
    return Q[:, :rank], rank


def reorder_by_interval(lam, vecs, Emin, Emax, M0):
    """Stable partition inside-first (src/core/feast_aux.jl:144-197).

    Returns (lam_new, vecs_new, ninside, perm) -- perm is 0-based.
    """
    inside = [i for i in range(M0) if Emin <= lam[i] <= Emax]
    outside = [i for i in range(M0) if not (Emin <= lam[i] <= Emax)]
    perm = np.array(inside + outside, dtype=np.int64)
    lam = np.array(lam, copy=True)
    vecs = np.array(vecs, copy=True)
    lam[:M0] = lam[perm]
    vecs[:, :M0] = vecs[:, perm]
    return lam, vecs, len(inside), perm


def inside_gcontour(lam, Emid, r, fpm18=100, fpm19=0):
    """src/core/feast_tools.jl:623-650."""
    w = complex(lam) - complex(Emid)
    aspect = fpm18 * 0.01 if fpm18 > 0 else 1.0
    if fpm19 != 0:
        w *= np.exp(-1j * (fpm19 / 180.0) * math.pi)
    x = w.real / r
    y = w.imag / (r * aspect)
    return x * x + y * y <= 1.0


def reorder_by_gcontour(lam, vecs, Emid, r, M0, fpm18=100, fpm19=0):
    """src/core/feast_aux.jl:208-257."""
    ins = [i for i in range(M0) if inside_gcontour(lam[i], Emid, r, fpm18, fpm19)]
    out = [i for i in range(M0) if not inside_gcontour(lam[i], Emid, r, fpm18, fpm19)]
    perm = np.array(ins + out, dtype=np.int64)
    lam = np.array(lam, copy=True)
    vecs = np.array(vecs, copy=True)
    lam[:M0] = lam[perm]
    vecs[:, :M0] = vecs[:, perm]
    return lam, vecs, len(ins), perm


def feast_sort(lam, q, res, M):
    """Stable insertion sort by eigenvalue (src/core/feast_tools.jl:653-682)."""
    order = sorted(range(M), key=lambda i: lam[i])  # python sort is stable
    lam = np.array(lam, copy=True); q = np.array(q, copy=True); res = np.array(res, copy=True)
    lam[:M] = lam[order]; res[:M] = res[order]; q[:, :M] = q[:, order]
    return lam, q, res


def feast_sort_general(lam, q, res, M):
    """Stable insertion sort by |lambda|^2 (src/core/feast_tools.jl:685-713)."""
    order = sorted(range(M), key=lambda i: abs(lam[i]) ** 2)
    lam = np.array(lam, copy=True); q = np.array(q, copy=True); res = np.array(res, copy=True)
    lam[:M] = lam[order]; res[:M] = res[order]; q[:, :M] = q[:, order]
    return lam, q, res


def feast_residual(A, B, lam, q, M):
    """res_j = ||A q_j - lam_j B q_j|| / max(|lam_j|, 1)  (src/core/feast_tools.jl:726-755)."""
    res = np.zeros(M)
    for j in range(M):
        qj = q[:, j]
        r = A @ qj - lam[j] * (qj if B is None else B @ qj)
        res[j] = np.linalg.norm(r) / max(abs(lam[j]), 1.0)
    return res


def distribute_contour_points(ne, nw):
    """Contiguous block partition (src/parallel/feast_parallel.jl:433-447).

    Returns a list of nw lists of 0-based node indices.
    """
    per, rem = divmod(ne, nw)
    chunks, start = [], 0
    for i in range(nw):
        size = per + (1 if i < rem else 0)
        chunks.append(list(range(start, start + size)))
        start += size
    return chunks


# ---------------------------------------------------------------------------
# a6. restarted GMRES, one column at a time
#   src/sparse/feast_sparse.jl:164-203 (solve_shifted_iterative!)
#   Krylov.jl 0.10.1 gmres semantics: x0 = 0, no preconditioner,
#   stop when ||r_k|| <= atol + rtol*||r_0||, at most itmax inner iterations.
# ---------------------------------------------------------------------------
def gmres_restarted(matvec, b, rtol, atol, itmax, restart):
    n = b.shape[0]
    x = np.zeros(n, dtype=np.complex128)
    r = b.astype(np.complex128).copy()
    beta0 = np.linalg.norm(r)
    target = atol + rtol * beta0
    if beta0 <= target:
        return x, True, 0
    m = max(restart, 2)
    it = 0
    beta = beta0
    while it < itmax:
        V = np.zeros((n, m + 1), dtype=np.complex128)
        H = np.zeros((m + 1, m), dtype=np.complex128)
        cs = np.zeros(m, dtype=np.complex128); sn = np.zeros(m, dtype=np.complex128)
        g = np.zeros(m + 1, dtype=np.complex128)
        V[:, 0] = r / beta
        g[0] = beta
        k_used = 0
        solved = False
        for k in range(m):
            w = matvec(V[:, k])
            for i in range(k + 1):          # modified Gram-Schmidt
                H[i, k] = np.vdot(V[:, i], w)
                w = w - H[i, k] * V[:, i]
            H[k + 1, k] = np.linalg.norm(w)
            if H[k + 1, k] != 0:
                V[:, k + 1] = w / H[k + 1, k]
            for i in range(k):              # apply previous Givens rotations
                t = cs[i] * H[i, k] + sn[i] * H[i + 1, k]
                H[i + 1, k] = -np.conj(sn[i]) * H[i, k] + cs[i] * H[i + 1, k]
                H[i, k] = t
            a, bb = H[k, k], H[k + 1, k]
            denom = math.sqrt(abs(a) ** 2 + abs(bb) ** 2)
            if denom == 0:
                cs[k], sn[k] = 1.0, 0.0
            else:
                cs[k] = abs(a) / denom if a != 0 else 0.0
                sn[k] = (a / abs(a)) * np.conj(bb) / denom if a != 0 else 1.0
            H[k, k] = cs[k] * a + sn[k] * bb
            H[k + 1, k] = 0.0
            g[k + 1] = -np.conj(sn[k]) * g[k]
            g[k] = cs[k] * g[k]
            it += 1
            k_used = k + 1
            if abs(g[k + 1]) <= target:
                solved = True
                break
            if it >= itmax:
                break
        y = sla.solve_triangular(H[:k_used, :k_used], g[:k_used])
        x = x + V[:, :k_used] @ y
        r = b - matvec(x)
        beta = np.linalg.norm(r)
        if solved or beta <= target:
            return x, True, it
    return x, False, it


def solve_shifted_iterative(dest, rhs, shifted_matvec, tol, maxiter, restart):
    """Column-by-column GMRES + explicit residual check.

    src/sparse/feast_sparse.jl:164-203; dense analogue src/dense/feast_dense.jl:26-67.
    Returns (ok, total_inner_iterations).
    """
    total = 0
    for j in range(rhs.shape[1]):
        b = rhs[:, j]
        x, solved, it = gmres_restarted(shifted_matvec, b, tol, tol, maxiter, restart)
        total += it
        res_norm = np.linalg.norm(shifted_matvec(x) - b)
        limit = 10 * tol * max(np.linalg.norm(b), 1.0)
        if (not solved) or res_norm > limit:
            return False, total
        dest[:, j] = x
    return True, total


# ---------------------------------------------------------------------------
# Variant A: "QR + Rayleigh-Ritz" serial Hermitian driver
#   dense : src/dense/feast_dense.jl:78-351
#   sparse: src/sparse/feast_sparse.jl:246-499
# ---------------------------------------------------------------------------
def _is_sparse(M):
    return sp.issparse(M)


def _splu(S):
    """SuperLU with a symmetric-pattern minimum-degree ordering -- the closest analogue of
    UMFPACK's symmetric strategy that `lu(::SparseMatrixCSC)` picks for these matrices."""
    return spla.splu(sp.csc_matrix(S), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.1,
                     options=dict(SymmetricMode=True))


def feast_hermitian(A, B, Emin, Emax, M0, ne=8, fpm3=12, fpm4=20, fpm16=0, fpm18=100,
                    solver="direct", solver_tol=0.0, solver_maxiter=500, solver_restart=30,
                    Q0=None, seed=20260515, contour=None, collect=None, real_projection=False, sweep=None):
    """Variant A.  A (and B or None) dense ndarray or scipy sparse, Hermitian.

    real_projection=False is the reference as written (complex half-contour sum).  True
    takes Q_proj = Re(sum 2 w_e Y_e) -- the full-contour FEAST filter that the reference's
    real paths use (src/parallel/feast_parallel.jl:38-55, src/kernel/feast_kernel.jl:183-186);
    valid for real-symmetric A, B with a real start.  The reference's variant A on a random
    start often ends with M=0 / info=5 at loop 0 on large problems (its half-contour filter
    decays like 1/distance), so this switch provides the working CPU answer for cfg 3.

    ``collect``: optional dict; when given, per-loop intermediates (Q_proj,
    rank, lambda, epsout) are appended for golden-vector generation.

    ``sweep``: optional callable(Q[:, :active]) -> sum_e 2 w_e (z_e B - A)^{-1} B Q that replaces the serial node
    loop below -- oracle/node_farm.py runs the nodes on host processes, the shape of the reference's :threads /
    :distributed backends (src/parallel/feast_parallel.jl:586-630, 484-503); same arithmetic per node.
    """
    N = A.shape[0]
    if N <= 0:
        return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), FEAST_ERROR_N, math.inf, 0)
    if M0 <= 0 or M0 > N:
        return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), FEAST_ERROR_M0, math.inf, 0)
    if not Emin < Emax:
        return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), FEAST_ERROR_EMIN_EMAX, math.inf, 0)
    sparse = _is_sparse(A)
    if sparse:
        A = sp.csc_matrix(A, dtype=np.complex128)
        Bc = None if B is None else sp.csc_matrix(B, dtype=np.complex128)
    else:
        A = np.asarray(A, dtype=np.complex128)
        Bc = None if B is None else np.asarray(B, dtype=np.complex128)
    tol_value = feast_tolerance(fpm3) if solver_tol == 0.0 else float(solver_tol)
    direct = solver == "direct"

    Q_basis = seeded_subspace(N, M0, seed) if Q0 is None else np.array(Q0, dtype=np.complex128, order="F")
    if contour is None:
        Zne, Wne = feast_contour(Emin, Emax, ne, fpm16, fpm18)
    else:
        Zne, Wne = contour
    factor_cache = [None] * len(Zne)
    eps_tol = feast_tolerance(fpm3)
    epsout = math.inf
    info = FEAST_SUCCESS
    loop_count = 0
    M_found = 0
    active = M0
    lam_vec = np.zeros(M0)
    res_vec = np.zeros(M0)
    solutions = np.zeros((N, M0), dtype=np.complex128, order="F")
    stats = {"inner_iterations": 0, "factorizations": 0}

    for loop_idx in range(0, fpm4 + 1):
        loop_count = loop_idx
        Q_proj = np.zeros((N, M0), dtype=np.complex128, order="F")
        failed = False
        if sweep is not None:
            try:
                Q_proj[:, :active] = sweep(Q_basis[:, :active])
            except Exception:
                info = FEAST_ERROR_LAPACK
                failed = True
        for e, z in enumerate(Zne if sweep is None else ()):
            weight = 2 * Wne[e]
            basis = Q_basis[:, :active]
            rhs = basis.copy() if Bc is None else Bc @ basis
            if direct:
                try:
                    if factor_cache[e] is None:
                        if sparse:
                            S = (z * sp.identity(N, dtype=np.complex128, format="csc") - A) if Bc is None else (z * Bc - A)
                            factor_cache[e] = _splu(S)
                        else:
                            S = dense_shifted_identity_minus(z, A) if Bc is None else z * Bc - A
                            factor_cache[e] = sla.lu_factor(S)
                        stats["factorizations"] += 1
                    if sparse:
                        Y = factor_cache[e].solve(np.ascontiguousarray(rhs))
                    else:
                        Y = sla.lu_solve(factor_cache[e], rhs)
                    if not np.all(np.isfinite(Y)):
                        raise np.linalg.LinAlgError("singular shifted system")
                except Exception:
                    info = FEAST_ERROR_LAPACK
                    failed = True
                    break
            else:
                if Bc is None:
                    mv = lambda x, z=z: z * x - A @ x
                else:
                    mv = lambda x, z=z: z * (Bc @ x) - A @ x
                Y = np.zeros_like(rhs)
                ok, its = solve_shifted_iterative(Y, rhs, mv, tol_value, solver_maxiter, solver_restart)
                stats["inner_iterations"] += its
                if not ok:
                    info = FEAST_ERROR_NO_CONVERGENCE
                    failed = True
                    break
            Q_proj[:, :active] += weight * Y
        if failed:
            break
        if real_projection:
            Q_proj = np.asfortranarray(Q_proj.real.astype(np.complex128))

        q_rank, rank = qr_compress(Q_proj, active)
        if rank == 0:
            info = FEAST_ERROR_NO_CONVERGENCE
            break
        Sq = hermitian_part(q_rank.conj().T @ (A @ q_rank))
        if Bc is None:
            Aq = np.eye(rank, dtype=np.complex128)
        else:
            Aq = hermitian_part(q_rank.conj().T @ (Bc @ q_rank))
        try:
            lam_red, v_red = sla.eigh(Sq, Aq)
        except Exception:
            try:
                w, v_red = sla.eig(Sq, Aq)
                lam_red = w.real
            except Exception:
                info = FEAST_ERROR_LAPACK
                break
        solutions[:, :rank] = q_rank @ v_red
        lam_vec[:rank] = lam_red
        lam_vec, solutions, M, _ = reorder_by_interval(lam_vec, solutions, Emin, Emax, rank)
        if collect is not None:
            collect.setdefault("loops", []).append(
                {"Q_proj": Q_proj.copy(), "rank": rank, "lambda": lam_vec[:rank].copy(), "M": M})
        if M == 0:
            info = FEAST_ERROR_NO_CONVERGENCE
            break
        for j in range(M):
            nrm = np.linalg.norm(solutions[:, j])
            if nrm > 0:
                solutions[:, j] /= nrm
        res_vec[:M] = feast_residual(A, Bc, lam_vec, solutions, M)
        epsout = float(res_vec[:M].max())
        M_found = M
        if epsout <= eps_tol:
            break
        if loop_idx == fpm4:
            info = FEAST_ERROR_NO_CONVERGENCE
            break
        active = rank
        Q_basis[:, :active] = solutions[:, :active]

    if M_found == 0 and info == FEAST_SUCCESS:
        info = FEAST_ERROR_NO_CONVERGENCE
    return FeastResult(lam_vec[:M_found].copy(), solutions[:, :M_found].copy(), M_found,
                       res_vec[:M_found].copy(), info, epsout, loop_count, stats)


def complex_to_real_result(res: FeastResult) -> FeastResult:
    """src/dense/feast_dense.jl:372-387 -- real symmetric callers get real.(q)."""
    return FeastResult(res.lam, np.real(res.q), res.M, res.res, res.info, res.epsout, res.loop, res.stats)


# ---------------------------------------------------------------------------
# Variant B: "moments"   (real symmetric, parallel drivers)
#   per-node worker: src/parallel/feast_parallel.jl:717-751 (sparse), :227-274 (dense)
#   outer loop     : src/parallel/feast_parallel.jl:450-572
# ---------------------------------------------------------------------------
def node_moments(work, Y, weight, z):
    """One quadrature node's contribution to the moment matrices (variant B):
        m_e = work^H Y_e ;  zAq += weight * m_e ;  zSq += weight * z_e * m_e
    src/kernel/feast_kernel.jl:146-153, 522-524 (RCI kernels), src/parallel/feast_mpi.jl:236-245, 564-567.
    Pinned by the reference's literal (test/test_allocation_helpers.jl:219-265): Aq = Wne[1]*(work'*workc),
    Bq = Zne[1]*Aq.  Returns (weight*m_e, weight*z*m_e)."""
    temp = work.conj().T @ Y
    return weight * temp, weight * z * temp


def pfeast_single_point(A, B, work, z, w, M0):
    """Returns (Aq, Sq, Q_proj) contribution of one node -- real parts, weight 2w."""
    N = A.shape[0]
    sparse = _is_sparse(A)
    W = work[:, :M0]
    if sparse:
        S = sp.csc_matrix(z * B - A, dtype=np.complex128)
        Y = spla.splu(S).solve(np.ascontiguousarray((B @ W).astype(np.complex128)))
    else:
        S = z * B - A
        Y = sla.lu_solve(sla.lu_factor(S), (B @ W).astype(np.complex128))
    weight = 2 * w
    a, s_ = node_moments(W, Y, weight, z)          # W is real here: W^H = W^T (feast_parallel.jl:742-747)
    return np.real(a), np.real(s_), np.real(weight * Y)


def pfeast_moments(A, B, Emin, Emax, M0, ne=8, fpm3=12, fpm4=20, nworkers=1, seed=20260515, Q0=None):
    """Variant B outer loop (src/parallel/feast_parallel.jl:450-572), real symmetric A, B."""
    N = A.shape[0]
    Zne, Wne = feast_contour(Emin, Emax, ne)
    work = np.real(seeded_subspace(N, M0, seed)) if Q0 is None else np.array(Q0, dtype=np.float64)
    eps_tol = feast_tolerance(fpm3)
    lam = np.zeros(M0); q = np.zeros((N, M0)); res = np.zeros(M0)
    chunks = distribute_contour_points(ne, nworkers)
    for loop in range(1, fpm4 + 1):
        Aq = np.zeros((M0, M0)); Sq = np.zeros((M0, M0)); Q_proj = np.zeros((N, M0))
        for chunk in chunks:
            for e in chunk:
                a, s, qp = pfeast_single_point(A, B, work, Zne[e], Wne[e], M0)
                Aq += a; Sq += s; Q_proj += qp
        try:
            # Symmetric(X) in Julia reads the upper triangle
            Su = np.triu(Sq) + np.triu(Sq, 1).T
            Au = np.triu(Aq) + np.triu(Aq, 1).T
            lam_red, v_red = sla.eigh(Su, Au)
        except Exception:
            w_, v_red = sla.eig(Sq, Aq)
            lam_red = w_.real; v_red = v_red.real
        q[:, :] = Q_proj @ v_red
        lam[:] = lam_red
        lam, q, M, _ = reorder_by_interval(lam, q, Emin, Emax, M0)
        if M == 0:
            return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), FEAST_ERROR_NO_CONVERGENCE, 0.0, loop)
        for j in range(M):
            n_ = np.linalg.norm(q[:, j])
            if n_ > 0:
                q[:, j] /= n_
        res[:M] = feast_residual(A, B, lam, q, M)
        epsout = float(res[:M].max())
        if epsout <= eps_tol:
            lam, q, res = feast_sort(lam, q, res, M)
            return FeastResult(lam[:M].copy(), q[:, :M].copy(), M, res[:M].copy(), FEAST_SUCCESS, epsout, loop)
        work[:, :M0] = q[:, :M0]
    M = sum(1 for i in range(M0) if Emin <= lam[i] <= Emax)
    eps_f = float(res[:M].max()) if M > 0 else 0.0
    return FeastResult(lam[:M].copy(), q[:, :M].copy(), M, res[:M].copy(), FEAST_ERROR_NO_CONVERGENCE, eps_f, fpm4)


def mpi_complex_hermitian(A, B, Emin, Emax, M0, ne=8, fpm3=12, fpm4=20, nworkers=1, seed=20260515, Q0=None):
    """Variant B for complex Hermitian input: _mpi_feast_complex_hermitian! (src/parallel/feast_mpi.jl:796-909) with
    the per-rank worker mpi_compute_complex_hermitian_moments (:523-571), direct solves.  ``nworkers`` only changes
    the order of the (exact) sums."""
    N = A.shape[0]
    sparse = _is_sparse(A)
    Ac = sp.csc_matrix(A, dtype=np.complex128) if sparse else np.asarray(A, dtype=np.complex128)
    if B is None:
        Bc = sp.identity(N, dtype=np.complex128, format="csc") if sparse else np.eye(N, dtype=np.complex128)
    else:
        Bc = sp.csc_matrix(B, dtype=np.complex128) if sparse else np.asarray(B, dtype=np.complex128)
    Zne, Wne = feast_contour(Emin, Emax, ne)
    Q = seeded_subspace(N, M0, seed, complex_values=True) if Q0 is None else np.array(Q0, dtype=np.complex128)
    eps_tol = feast_tolerance(fpm3)
    lam = np.zeros(M0); res = np.zeros(M0); X = np.zeros((N, M0), dtype=np.complex128)
    info, epsout, M_found, loop_count = FEAST_SUCCESS, math.inf, 0, 0
    chunks = distribute_contour_points(ne, nworkers)
    for loop_idx in range(0, fpm4 + 1):
        loop_count = loop_idx
        zAq = np.zeros((M0, M0), dtype=np.complex128); zSq = np.zeros_like(zAq); Q_proj = np.zeros((N, M0), dtype=np.complex128)
        rhs = Bc @ Q
        for chunk in chunks:
            for e in chunk:
                S = Zne[e] * Bc - Ac
                Y = spla.splu(sp.csc_matrix(S)).solve(np.ascontiguousarray(rhs)) if sparse else np.linalg.solve(S, rhs)
                weight = 2 * Wne[e]
                a, s_ = node_moments(Q, Y, weight, Zne[e])
                zAq += a; zSq += s_; Q_proj += weight * Y
        Aq, Sq = hermitian_part(zAq), hermitian_part(zSq)
        try:
            lam_red, v_red = sla.eigh(Sq, Aq)
        except Exception:
            w_, v_red = sla.eig(Sq, Aq)
            lam_red = w_.real
        X[:, :] = Q_proj @ v_red
        lam[:] = lam_red
        lam, X, M, _ = reorder_by_interval(lam, X, Emin, Emax, M0)
        if M == 0:
            info = FEAST_ERROR_NO_CONVERGENCE
            break
        for j in range(M):
            n_ = np.linalg.norm(X[:, j])
            if n_ > 0:
                X[:, j] /= n_
        res[:M] = feast_residual(Ac, Bc, lam, X, M)
        epsout = float(res[:M].max())
        M_found = M
        if epsout <= eps_tol:
            break
        if loop_idx == fpm4:
            info = FEAST_ERROR_NO_CONVERGENCE
            break
        Q = X.copy()
    if M_found > 1:
        lam, X, res = feast_sort(lam, X, res, M_found)
    return FeastResult(lam[:M_found].copy(), X[:, :M_found].copy(), M_found, res[:M_found].copy(), info, epsout, loop_count)


# ---------------------------------------------------------------------------
# Variant C: general two-sided RCI maths
#   kernel : src/kernel/feast_kernel.jl:646-962 (feast_grci!)
#   caller : src/dense/feast_dense.jl:402-593   (feast_gegv!)
# ---------------------------------------------------------------------------
def feast_general(A, B, Emid, r, M0, ne=16, fpm3=12, fpm4=20, fpm16=0, fpm18=100, fpm19=0,
                  Q0=None, seed=20260515):
    N = A.shape[0]
    if r <= 0:
        return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0), FEAST_ERROR_EMID_R, math.inf, 0)
    if M0 <= 0 or M0 > N:
        return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0), FEAST_ERROR_M0, math.inf, 0)
    sparse = _is_sparse(A)
    A = sp.csc_matrix(A, dtype=np.complex128) if sparse else np.asarray(A, dtype=np.complex128)
    if B is not None:
        B = sp.csc_matrix(B, dtype=np.complex128) if sparse else np.asarray(B, dtype=np.complex128)
    Zne, Wne = feast_gcontour(Emid, r, ne, fpm16, fpm18, fpm19)
    Q = seeded_subspace(N, M0, seed) if Q0 is None else np.array(Q0, dtype=np.complex128)
    eps_tol = feast_tolerance(fpm3)
    factors = {}
    lam = np.zeros(M0, dtype=np.complex128)
    res = np.zeros(M0)
    loop = 0
    while True:
        q = np.zeros((N, M0), dtype=np.complex128)
        for e, z in enumerate(Zne):
            if e not in factors:
                if sparse:
                    S = z * (sp.identity(N, dtype=np.complex128, format="csc") if B is None else B) - A
                    factors[e] = spla.splu(sp.csc_matrix(S))
                else:
                    S = dense_shifted_identity_minus(z, A) if B is None else z * B - A
                    factors[e] = sla.lu_factor(S)
            rhs = Q if B is None else B @ Q
            Y = factors[e].solve(np.ascontiguousarray(rhs)) if sparse else sla.lu_solve(factors[e], rhs)
            q += Wne[e] * Y                        # feast_kernel.jl:762-766 -- no factor 2
        BQ = q if B is None else B @ q
        Sq = q.conj().T @ BQ                       # :790
        Aq = q.conj().T @ (A @ q)                  # :805
        try:
            lam_red, v_red = sla.eig(Aq, Sq)       # :812
        except Exception:
            return FeastResult(lam[:0], q[:, :0], 0, res[:0], FEAST_ERROR_LAPACK, math.inf, loop)
        ins = [i for i in range(M0) if inside_gcontour(lam_red[i], Emid, r, fpm18, fpm19)]
        M = len(ins)
        if M == 0:
            return FeastResult(lam[:0], q[:, :0], 0, res[:0], FEAST_ERROR_NO_CONVERGENCE, math.inf, loop)
        out = [i for i in range(M0) if i not in set(ins)]
        perm = ins + out
        X = q @ v_red
        X = X[:, perm]
        lam = lam_red[perm]
        nrm = np.linalg.norm(X, axis=0)
        nrm[nrm == 0] = 1.0
        X = X / nrm
        AX = A @ X[:, :M]
        for j in range(M):                          # :899-906 -- residual WITHOUT B
            res[j] = np.linalg.norm(AX[:, j] - lam[j] * X[:, j]) / max(abs(lam[j]), 1.0)
        epsout = float(res[:M].max())
        if epsout <= eps_tol or loop >= fpm4:
            lam, X, res = feast_sort_general(lam, X, res, M)
            return FeastResult(lam[:M].copy(), X[:, :M].copy(), M, res[:M].copy(), FEAST_SUCCESS, epsout, loop)
        loop += 1
        Q = X


# ---------------------------------------------------------------------------
# Complex-symmetric sibling of variant A -- src/dense/feast_dense.jl:1026-1259,
# src/sparse/feast_sparse.jl:509-711
# ---------------------------------------------------------------------------
def feast_complex_symmetric(A, B, Emid, r, M0, ne=16, fpm3=12, fpm4=20, fpm16=0, fpm18=100, fpm19=0,
                            Q0=None, seed=20260515):
    """Full contour, no factor 2 (:1107), pivoted-QR compression (:1163), BILINEAR projection
    Ared = q^T A q, Bred = q^T B q (:1181-1182), eigen(Ared, Bred) (:1184), inside-first reorder by
    the contour (:1193), ALL rank columns normalised (:1200-1209), residual WITH B (:1211-1224),
    final sort by |lambda|^2 (:1250)."""
    A = np.asarray(A.todense() if _is_sparse(A) else A, dtype=np.complex128)
    N = A.shape[0]
    if not np.array_equal(A, A.T):
        raise ValueError("Matrix A must be complex symmetric (A == transpose(A))")
    Bd = None if B is None else np.asarray(B.todense() if _is_sparse(B) else B, dtype=np.complex128)
    if Bd is not None and not np.array_equal(Bd, Bd.T):
        raise ValueError("Matrix B must be complex symmetric (B == transpose(B))")
    Zne, Wne = feast_gcontour(Emid, r, ne, fpm16, fpm18, fpm19)
    Q = seeded_subspace(N, M0, seed, complex_values=True) if Q0 is None else np.array(Q0, dtype=np.complex128)
    eps_tol = feast_tolerance(fpm3)
    factors = {}
    lam = np.zeros(M0, dtype=np.complex128)
    X = np.zeros((N, M0), dtype=np.complex128)
    res = np.zeros(M0)
    info, epsout, M_found, active, loop_count = FEAST_SUCCESS, math.inf, 0, M0, 0
    for loop in range(0, fpm4 + 1):
        loop_count = loop
        Qp = np.zeros((N, active), dtype=np.complex128)
        rhs = Q[:, :active] if Bd is None else Bd @ Q[:, :active]
        for e, z in enumerate(Zne):
            if e not in factors:
                factors[e] = sla.lu_factor(dense_shifted_identity_minus(z, A) if Bd is None else z * Bd - A)
            Qp += Wne[e] * sla.lu_solve(factors[e], rhs)
        q, rank = qr_compress(Qp, active)
        if rank == 0:
            info = FEAST_ERROR_NO_CONVERGENCE
            break
        Ared = q.T @ (A @ q)
        Bred = q.T @ (q if Bd is None else Bd @ q)
        try:
            lam_red, v_red = sla.eig(Ared, Bred)
        except Exception:
            info = FEAST_ERROR_LAPACK
            break
        Xr = q @ v_red
        ins = [i for i in range(rank) if inside_gcontour(lam_red[i], Emid, r, fpm18, fpm19)]
        perm = ins + [i for i in range(rank) if i not in set(ins)]
        M = len(ins)
        if M == 0:
            info = FEAST_ERROR_NO_CONVERGENCE
            break
        lam[:rank] = lam_red[perm]
        Xr = Xr[:, perm]
        nrm = np.linalg.norm(Xr, axis=0)
        nrm[nrm == 0] = 1.0
        Xr = Xr / nrm
        X[:, :rank] = Xr
        for j in range(M):
            xj = Xr[:, j]
            res[j] = np.linalg.norm(A @ xj - lam[j] * (xj if Bd is None else Bd @ xj)) / max(abs(lam[j]), 1.0)
        epsout = float(res[:M].max())
        M_found = M
        if epsout <= eps_tol:
            break
        if loop == fpm4:
            info = FEAST_ERROR_NO_CONVERGENCE
            break
        active = rank
        Q = Xr.copy()
    if M_found == 0 and info == FEAST_SUCCESS:
        info = FEAST_ERROR_NO_CONVERGENCE
    if M_found > 1:
        lam, X, res = feast_sort_general(lam, X, res, M_found)
    return FeastResult(lam[:M_found].copy(), X[:, :M_found].copy(), M_found, res[:M_found].copy(), info, epsout, loop_count)


# ---------------------------------------------------------------------------
# RCI kernels, jobs served exactly -- src/kernel/feast_kernel.jl:7-275 (srci), :397-644 (hrci)
# (the general kernel grci, :646-962, is feast_general above)
# ---------------------------------------------------------------------------
def _ggev_scaled(S, A_):
    """eigen(S, A) as LAPACK ggev leaves it (Julia does not rescale): each eigenvector has
    max_i |re v_i| + |im v_i| = 1.  scipy normalises to unit 2-norm, which is undone here; the
    srci/hrci residuals are taken on un-normalised Ritz vectors and depend on this."""
    w, V = sla.eig(S, A_)
    V = np.array(V, dtype=np.complex128)
    for j in range(V.shape[1]):
        s = np.max(np.abs(V[:, j].real) + np.abs(V[:, j].imag))
        if s > 0:
            V[:, j] = V[:, j] / s
    return w, V


def rci_symmetric(A, B, Emin, Emax, M0, ne=8, fpm3=12, fpm4=20, Q0=None, seed=20260515, contour=None,
                  rhs_uses_B=True):
    """What feast_srci! returns when its FACTORIZE/SOLVE/MULT_A jobs are served with exact dense
    solves (caller loops: src/banded/feast_banded.jl:87-175 -- rhs = B*work; matrix-free
    src/interfaces/feast_matfree.jl:203-254 -- rhs = work, ``rhs_uses_B=False``).
    Moments in complex, real part after the sweep (:146-169); reduced pencil eigen(Sq, Aq) (:175);
    q = Re(Q_proj) V with no normalisation (:183-187); inside-first stable reorder (:189-215);
    residual ||A q - lambda q|| / max(|lambda|,1) WITHOUT B (:244-252); stop test uses loop >=
    fpm[4] (:258); all M0 Ritz vectors are the next trial subspace (:269)."""
    A = np.asarray(A.todense() if _is_sparse(A) else A, dtype=np.float64)
    N = A.shape[0]
    Bd = np.eye(N) if B is None else np.asarray(B.todense() if _is_sparse(B) else B, dtype=np.float64)
    Zne, Wne = contour if contour is not None else feast_contour(Emin, Emax, ne)
    if Q0 is None:
        Q = np.real(seeded_subspace(N, M0, seed))
    else:
        Q = np.array(Q0, dtype=np.float64)
        Q = Q / np.linalg.norm(Q, axis=0)
    eps_tol = feast_tolerance(fpm3)
    loop = 0
    while True:
        Qp = np.zeros((N, M0), dtype=np.complex128)
        zA = np.zeros((M0, M0), dtype=np.complex128)
        zS = np.zeros((M0, M0), dtype=np.complex128)
        for e in range(len(Zne)):
            Y = np.linalg.solve(Zne[e] * Bd - A, (Bd @ Q if rhs_uses_B else Q).astype(np.complex128))
            wt = 2 * Wne[e]
            Qp += wt * Y
            mom = Q.T @ Y
            zA += wt * mom
            zS += Zne[e] * (wt * mom)
        try:
            w, V = _ggev_scaled(zS.real, zA.real)
        except Exception:
            return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), FEAST_ERROR_LAPACK, 0.0, loop)
        lam = np.real(w)
        X = Qp.real @ np.real(V)
        ins = [i for i in range(M0) if Emin <= lam[i] <= Emax]
        perm = ins + [i for i in range(M0) if not (Emin <= lam[i] <= Emax)]
        M = len(ins)
        lam, X = lam[perm], X[:, perm]
        if M == 0:
            return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), FEAST_ERROR_NO_CONVERGENCE, 0.0, loop)
        res = np.zeros(M0)
        AX = A @ X[:, :M]
        for j in range(M):
            res[j] = np.linalg.norm(AX[:, j] - lam[j] * X[:, j]) / max(abs(lam[j]), 1.0)
        epsout = float(res[:M].max())
        if epsout <= eps_tol or loop >= fpm4:
            lam, X, res = feast_sort(lam, X, res, M)
            return FeastResult(lam[:M].copy(), X[:, :M].copy(), M, res[:M].copy(), FEAST_SUCCESS, epsout, loop)
        loop += 1
        Q = X.copy()


def rci_hermitian(A, B, Emin, Emax, M0, ne=8, fpm3=12, fpm4=20, Q0=None, seed=20260515, contour=None):
    """What feast_hrci! returns with exact solves (feast_kernel.jl:397-644): complex trial
    subspace, zAq/zSq accumulated in place WITHOUT a per-sweep reset other than the one at the
    start of a refinement loop (:618-619), reduced pencil eigen(zSq, zAq) in complex (:539),
    q = Q_proj V complex (:547), lambda = real parts (:540)."""
    A = np.asarray(A.todense() if _is_sparse(A) else A, dtype=np.complex128)
    N = A.shape[0]
    Bd = np.eye(N, dtype=np.complex128) if B is None else np.asarray(B.todense() if _is_sparse(B) else B, dtype=np.complex128)
    Zne, Wne = contour if contour is not None else feast_contour(Emin, Emax, ne)
    if Q0 is None:
        Q = seeded_subspace(N, M0, seed, complex_values=True)
    else:
        Q = np.array(Q0, dtype=np.complex128)
        Q = Q / np.linalg.norm(Q, axis=0)
    eps_tol = feast_tolerance(fpm3)
    loop = 0
    while True:
        Qp = np.zeros((N, M0), dtype=np.complex128)
        zA = np.zeros((M0, M0), dtype=np.complex128)
        zS = np.zeros((M0, M0), dtype=np.complex128)
        for e in range(len(Zne)):
            Y = np.linalg.solve(Zne[e] * Bd - A, Bd @ Q)
            wt = 2 * Wne[e]
            Qp += wt * Y
            a, s_ = node_moments(Q, Y, wt, Zne[e])
            zA += a
            zS += s_
        try:
            w, V = _ggev_scaled(zS, zA)
        except Exception:
            return FeastResult(np.zeros(0), np.zeros((N, 0), complex), 0, np.zeros(0), FEAST_ERROR_LAPACK, 0.0, loop)
        lam = np.real(w)
        X = Qp @ V
        ins = [i for i in range(M0) if Emin <= lam[i] <= Emax]
        perm = ins + [i for i in range(M0) if not (Emin <= lam[i] <= Emax)]
        M = len(ins)
        lam, X = lam[perm], X[:, perm]
        if M == 0:
            return FeastResult(np.zeros(0), np.zeros((N, 0), complex), 0, np.zeros(0), FEAST_ERROR_NO_CONVERGENCE, 0.0, loop)
        res = np.zeros(M0)
        AX = A @ X[:, :M]
        for j in range(M):
            res[j] = np.linalg.norm(AX[:, j] - lam[j] * X[:, j]) / max(abs(lam[j]), 1.0)
        epsout = float(res[:M].max())
        if epsout <= eps_tol or loop >= fpm4:
            lam, X, res = feast_sort(lam, X, res, M)
            return FeastResult(lam[:M].copy(), X[:, :M].copy(), M, res[:M].copy(), FEAST_SUCCESS, epsout, loop)
        loop += 1
        Q = X.copy()


# ---------------------------------------------------------------------------
# Synthetic BASELINE inputs with closed-form spectra
# ---------------------------------------------------------------------------
def laplacian_3d(nx, ny, nz):
    """7-point Dirichlet Laplacian, lexicographic x-fastest, CSR float64."""
    def t(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")
    Ix, Iy, Iz = sp.identity(nx), sp.identity(ny), sp.identity(nz)
    A = sp.kron(Iz, sp.kron(Iy, t(nx))) + sp.kron(Iz, sp.kron(t(ny), Ix)) + sp.kron(t(nz), sp.kron(Iy, Ix))
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


def laplacian_3d_eigs(nx, ny, nz):
    mx = 2 - 2 * np.cos(np.arange(1, nx + 1) * np.pi / (nx + 1))
    my = 2 - 2 * np.cos(np.arange(1, ny + 1) * np.pi / (ny + 1))
    mz = 2 - 2 * np.cos(np.arange(1, nz + 1) * np.pi / (nz + 1))
    return np.sort((mx[:, None, None] + my[None, :, None] + mz[None, None, :]).ravel())


def cfg3_problem(nx=50, ny=40, nz=25, beta=0.1):
    """A = 3-D Laplacian, B = I + beta*A; lambda = mu/(1+beta*mu)."""
    A = laplacian_3d(nx, ny, nz)
    B = sp.csr_matrix(sp.identity(A.shape[0], format="csr") + beta * A)
    B.sort_indices()
    mu = laplacian_3d_eigs(nx, ny, nz)
    return A, B, np.sort(mu / (1 + beta * mu))


def householder_conjugated_diag(d, seed=20260515, nreflect=2, dtype=np.float64):
    """A = H2 H1 diag(d) H1 H2 with seeded unit reflectors (cfg 2)."""
    n = d.shape[0]
    rng = np.random.default_rng(seed)
    A = np.diag(d.astype(dtype))
    for _ in range(nreflect):
        v = rng.standard_normal(n)
        if np.issubdtype(dtype, np.complexfloating):
            v = v + 1j * rng.standard_normal(n)
        v = v / np.linalg.norm(v)
        # A <- H A H, H = I - 2 v v^H
        Av = A @ v
        A = A - 2 * np.outer(Av, v.conj())
        vA = v.conj() @ A
        A = A - 2 * np.outer(v, vA)
    if not np.issubdtype(dtype, np.complexfloating):
        A = 0.5 * (A + A.T)
    return A
