"""User-facing entry points with the reference's names and argument meaning
(src/interfaces/feast_interfaces.jl:143-379), restricted to the ``:hip`` backend this
package provides.  Everything else of the reference API stays on the Julia host."""
from __future__ import annotations

import math
import os
import warnings

import numpy as np
import scipy.sparse as sp

from .engine import HipEngine
from .hip_backend import feast_hip_general, feast_hip_hermitian
from .parameters import feastdefault, feastinit
from .types import FEAST_UNINITIALIZED, FeastHipError, FeastResult

_BACKENDS = ("hip", "auto")   # _normalize_backend whitelist edit, feast_interfaces.jl:44


def _is_hermitian(A, tol=0.0):
    if sp.issparse(A):
        D = (A - A.conj().T)
        return D.nnz == 0 or abs(D).max() <= tol
    return np.array_equal(A, A.conj().T)


def _sparse_direct_solver(A, B, nodes, budget_bytes=32 << 30, dense_limit=12288):
    """``solver=:direct`` for sparse input (UMFPACK in the reference, src/sparse/feast_sparse.jl:334-342, :943):
      "banded" -- the union pattern of A and B is a narrow band whose LU factors (2 kl + ku + 1 rows per column and
                  node, complex128) fit the budget: batched banded LU on the CSR arrays;
      "dense"  -- any other pattern up to N = dense_limit whose N x N complex128 factors (one per node, plus A and B)
                  fit the budget: the matrix is expanded on ingest and factored by the batched dense LU -- an exact
                  direct solve like the reference's, at O(N^3) per node, which the MFMA LU affords at these sizes;
      "krylov" -- everything larger: the batched Krylov solvers (north_star's iterative path)."""
    kl = ku = 0
    for M in (A, B):
        if M is None:
            continue
        c = sp.coo_matrix(M)
        if c.nnz:
            kl = max(kl, int((c.row - c.col).max()))
            ku = max(ku, int((c.col - c.row).max()))
    ldab = 2 * kl + ku + 1
    N = A.shape[0]
    if 2 * kl + ku + 1 <= 3500 and ldab * N * 16 * max(nodes, 1) <= budget_bytes and kl + ku <= 512:
        return "banded"
    if N <= dense_limit and (max(nodes, 1) + 2) * N * N * 16 <= budget_bytes:
        return "dense"
    return "krylov"


# `solver=:direct` on a sparse Hermitian pencil beyond the narrow-band / dense windows: the sparse direct solver (multifrontal
# LU, or the band LU) is taken outright when all its factorisations together cost at most this many flop (feasthip_direct_plan_flops
# x local nodes), else the Krylov path runs first with the direct solver as its fallback.  4e11: with the multifrontal plan a 2-D
# pencil of 50 000 unknowns (1e11 in all) is a 0.1 s direct call against 0.5 s of Krylov loops, while cfg 3 (1.7e12 in all: 0.19 s)
# stays with the Krylov path (0.16 s).  With the band LU alone (round 3) the switch was off: its block steps are a chain of panel
# kernels, ~0.4 s for N = 50 000 however thin the band.  0: never.
_DIRECT_FLOPS = float(os.environ.get("FEASTKIT_DIRECT_FLOPS", "4e11"))


def _direct_label(eng):
    """what FEASTHIP_SOLVER_BANDED runs for the problem set on the engine (feasthip_band_plan: 2 = multifrontal)."""
    try:
        return "multifrontal LU on a nested-dissection tree" if eng.band_plan()[3] == 2 else "band LU after reverse Cuthill-McKee"
    except Exception:
        return "band LU after reverse Cuthill-McKee"


def _band_direct_fits(eng, A, B, nodes, group=None, complexify=False, max_flops=1e14):
    """True when the sparse direct solver for general patterns (multifrontal LU on a nested-dissection tree, or reverse
    Cuthill-McKee + blocked band LU on the dense kernels: FEASTHIP_SOLVER_BANDED) can hold one factor per local quadrature node in the free device memory.  Sets the
    problem on the engine (a later set_problem with the same matrices is free: content fingerprint)."""
    import torch
    if complexify:
        A = A.astype(np.complex128) if not np.iscomplexobj(A.data) else A
        B = None if B is None else (B.astype(np.complex128) if not np.iscomplexobj(B.data) else B)
    world = 1
    if group is not None:
        import torch.distributed as dist
        world = dist.get_world_size(group)
    local = -(-max(int(nodes), 1) // world)
    try:
        eng.set_problem(A, B)
        kl, ku, nbytes, blocked = eng.band_plan()
        free, _total = torch.cuda.mem_get_info(eng.device)
        panels = 2 * local * A.shape[0] * 64 * 16
        # the elimination costs 8 N kl (kl + ku) flop per node: beyond ~1e14 in all (seconds of MFMA time) a band this wide
        # is no longer the cheap way to a direct solve
        # (or the multifrontal elimination's padded fronts, when the library's plan took that: blocked == 2)
        flops = eng.direct_plan_flops() * local
        fits = bool((local + 1) * nbytes + panels <= 0.85 * free and flops <= max_flops)
    except FeastHipError:
        fits = False
    if world > 1:
        # every rank decides from ITS free memory: the ranks must agree (different solvers have different tolerance
        # semantics, and a rank that alone takes the band LU would fail later in its slot allocation) -- logical AND over
        # the group, through the control plane (any backend)
        votes = [None] * world
        dist.all_gather_object(votes, fits, group=group)
        fits = all(votes)
    return fits


def _single_precision(*mats):
    """True when every matrix given is float32 / complex64: the reference preserves the element type of such input
    (test/runtests.jl:281-304) and stops at max(10^-fpm[3], sqrt(eps(Float32))) (src/core/feast_parameters.jl:398-405)."""
    dts = [np.dtype((M.dtype if not sp.issparse(M) else M.data.dtype)) for M in mats if M is not None]
    return bool(dts) and all(dt in (np.dtype(np.float32), np.dtype(np.complex64)) for dt in dts)


def _promote(M):
    """float32 / complex64 -> float64 / complex128 (the C ABI takes f64 / c128 only: the shim converts on the way in and
    casts the results back, so that single-precision callers keep their element types)."""
    if M is None:
        return None
    dt = np.complex128 if np.iscomplexobj(M.data if sp.issparse(M) else M) else np.float64
    return M.astype(dt)


def _demote(res, cplx_lambda):
    return FeastResult(res.lambda_.astype(np.complex64 if cplx_lambda else np.float32),
                       res.q.astype(np.complex64 if np.iscomplexobj(res.q) else np.float32), res.M, res.res.astype(np.float32), res.info,
                       float(res.epsout), res.loop, res.stats)


def _densify(M):
    return None if M is None else np.asfortranarray(M.toarray())


# What `solver=:direct` (the reference's default, UMFPACK for sparse input: src/sparse/feast_sparse.jl:334-342 via
# src/core/feast_backend_utils.jl:166-198) maps to when the pattern is not a narrow band: the batched Krylov path in
# the configuration that converges on large problems -- COCG for real-symmetric input (z B - A is complex symmetric:
# one operator application per iteration), BiCGStab otherwise; Ritz-pair warm starts, inexact inner solves (relative
# reduction 3e-2 per refinement loop, at most 100 iterations per loop) and, for real input, the real projection.
# The outer stop test is the reference's (max residual of the pairs inside <= 10^-fpm[3]).
_KRYLOV_DEFAULT = dict(warm_start=True, inner_rtol=3e-2, solver_maxiter=100)


_warned = set()


def _warn_substitution(sub):
    """One warning per process and configuration: the reference's `solver=:direct` means a sparse LU, which this
    backend replaces by a Krylov solver for patterns that are not a narrow band (also recorded in
    ``result.stats["solver_substitution"]``)."""
    key = (sub["used"], sub["warm_start"], sub["inner_rtol"], sub["solver_maxiter"])
    if key in _warned:
        return
    _warned.add(key)
    warnings.warn("feastkit.jl_amd: solver='direct' on a sparse pattern that is not a narrow band runs the batched "
                  f"{sub['used']} solver (warm_start={sub['warm_start']}, inner_rtol={sub['inner_rtol']}, "
                  f"<= {sub['solver_maxiter']} iterations per refinement loop) instead of a sparse LU",
                  RuntimeWarning, stacklevel=3)


def _release_band_factors(eng, keep):
    """The sparse direct solver's factors stay on the engine across set_contour and across calls (16 x 2.8 GB on cfg 3).
    Inside one call that is the reference's factor cache; beyond it, it is device memory a later Krylov / GMRES call on
    the same engine no longer has (and that _band_direct_fits would count as taken).  Whatever path of the call created
    them -- solver='banded', the automatic choice of feast_general, the Krylov-to-direct fallback -- they go when the
    call returns unless the caller asked to keep them."""
    if keep:
        return
    try:
        eng.free_factors()
    except FeastHipError:
        pass                                   # a poisoned handle reports through the result, not from the clean-up


def _engine(engine, device):
    return engine if engine is not None else HipEngine(device)


def feast(A, B=None, interval=None, *, M0=10, fpm=None, backend="hip", solver="direct", solver_tol=0.0,
          solver_maxiter=None, solver_restart=30, warm_start=None, inner_rtol=None, real_projection=None,
          inner_precision=64, group=None, engine=None, device=0, Q0=None, contour=None, contour_policy=None,
          keep_factors=False):
    """feast(A, [B,] (Emin, Emax); M0, fpm, backend=:hip) for real-symmetric / Hermitian
    dense (numpy) or sparse (scipy) matrices.  Real input is complexified and the result is
    real.(q), exactly as feast_sygv!/feast_scsrgv! do (src/dense/feast_dense.jl:362-387).
    ``contour=(Zne, Wne)``: caller-supplied half-contour nodes and weights, the reference's "x" drivers
    (feast_hcsrgvx!/feast_heevx!, test/runtests.jl:415-440).
    ``keep_factors``: the band-LU factors of the sparse direct solver (one per quadrature node: 2.8 GB each on cfg 3) are
    cached across the refinement loops of ONE call, like the reference's per-call ``lu(zB - A)`` cache
    (src/sparse/feast_sparse.jl:335-341), and released when the call returns; ``True`` leaves them resident on the engine
    for a repeated call on the same contour (``engine.free_factors()`` releases them).
    """
    if interval is None and B is not None and isinstance(B, tuple):
        B, interval = None, B              # feast(A, (Emin, Emax)) form
    if backend not in _BACKENDS:
        raise ValueError(f"Unknown backend '{backend}' (this package provides: hip)")
    if A.shape[0] != A.shape[1]:
        raise ValueError("Matrix A must be square")
    if B is not None and B.shape != A.shape:
        raise ValueError("Matrix B must match size of A")
    eng = _engine(engine, device)
    # input checks and the pattern scan cost tens of milliseconds on a 50 000-unknown CSR pair (sparse transposes): a
    # repeated call with the same matrices (content fingerprint, engine.py) skips them
    fp = (HipEngine._fingerprint(A), HipEngine._fingerprint(B)) if sp.issparse(A) else None
    known = fp is not None and fp[0] and fp[1] is not False and getattr(eng, "_checked", {}).get("fp") == fp
    if not known:
        if not _is_hermitian(A):
            raise ValueError("Matrix A must be Hermitian")
        if B is not None and not _is_hermitian(B):
            raise ValueError("Matrix B must be Hermitian positive definite")
    single = _single_precision(A, B)
    if single:
        A, B = _promote(A), _promote(B)
    Emin, Emax = float(interval[0]), float(interval[1])
    fpm = feastinit() if fpm is None else fpm
    aspect_unset = int(fpm[18]) == FEAST_UNINITIALIZED      # the caller left the contour shape to the library
    feastdefault(fpm)
    N = A.shape[0]
    M0 = min(int(M0), N)
    real_input = not (np.iscomplexobj(A.data if sp.issparse(A) else A) or
                      (B is not None and np.iscomplexobj(B.data if sp.issparse(B) else B)))
    substituted = None
    if sp.issparse(A) and solver in ("direct", "lu"):
        # the reference's sparse default is UMFPACK; the :hip backend has a direct path for band
        # matrices (batched banded LU) and otherwise the batched Krylov solver (north_star)
        if known and eng._checked.get("nodes") == int(fpm[2]):
            solver = eng._checked["direct"]
        else:
            solver = _sparse_direct_solver(A, B, int(fpm[2]))
            if fp is not None and fp[0] and fp[1] is not False:
                eng._checked = {"fp": fp, "nodes": int(fpm[2]), "direct": solver}
        if solver == "dense":
            A, B, solver = _densify(A), _densify(B), "direct"
            substituted = {"requested": "direct", "used": "dense LU of the expanded matrix"}
        elif (solver == "krylov" and group is None and _DIRECT_FLOPS > 0 and A.shape[0] <= 400_000     # (beyond: the plan itself -- host nested dissection -- is no longer small change)
              and _band_direct_fits(eng, A, B, int(fpm[2]), max_flops=_DIRECT_FLOPS)):
            # a band the direct solver eliminates in a fraction of a second (2-D problems, thin 3-D ones): the reference's own
            # default -- a direct factorisation per node, exact solves, two or three loops -- is then also the faster one
            # (measured: DESIGN.md section 5); wider bands keep the Krylov fast path with the direct solver as its fallback
            solver = "banded"
            substituted = {"requested": "direct", "used": _direct_label(eng)}
            if fp is not None and fp[0] and fp[1] is not False:
                eng._checked = {"fp": fp, "nodes": int(fpm[2]), "direct": solver}
        elif solver == "krylov":
            solver = "cocg" if real_input else "bicgstab"
            # each default applies on its own: naming one of the three keeps the other two
            if warm_start is None:
                warm_start = _KRYLOV_DEFAULT["warm_start"]
            if inner_rtol is None and warm_start:
                inner_rtol = _KRYLOV_DEFAULT["inner_rtol"]
            # inexact solves pay for a sharper filter than they can use: unless the caller fixed fpm[18], the driver
            # picks the ellipse ratio itself (hip_backend.feast_hip_hermitian, contour_policy)
            auto = contour_policy is None and aspect_unset and contour is None and warm_start and inner_rtol is not None
            if solver_maxiter is None:
                # per-loop iteration cap: 100 on the caller's contour; 50 under the contour policy, whose taller ellipses
                # need fewer iterations and whose safeguard doubles the cap when a loop stalls on capped nodes (measured on
                # four 50 000-unknown pencils: 50 is 5-10 % ahead on three and 5 % behind on the fourth)
                solver_maxiter = (50 if auto else _KRYLOV_DEFAULT["solver_maxiter"]) if (warm_start and inner_rtol is not None) else 500
            substituted = {"requested": "direct", "used": solver, "warm_start": bool(warm_start),
                           "inner_rtol": inner_rtol, "solver_maxiter": int(solver_maxiter)}
            _warn_substitution(substituted)
            if auto:
                contour_policy = "auto"
                substituted["contour_policy"] = "auto"
    if solver == "sparse_direct":
        solver = "banded"
    abort_check = None
    if (substituted is not None and substituted.get("used") in ("cocg", "bicgstab") and group is None
            and os.environ.get("FEASTKIT_DIRECT_SWITCH", "1") != "0"):
        # (FEASTKIT_DIRECT_SWITCH=0: never leave the Krylov path before fpm[4] loops are spent.)  The Krylov path was OUR choice for the caller's solver=:direct.  After every refinement loop the time it still needs
        # is extrapolated from the contraction of the outer residual over the last two loops; when that exceeds 3 x what
        # the sparse direct solver would take from here (its factorisations: a chain of ~1.5 ms block steps plus
        # 8 N kl (kl + ku) flop per node at ~20 TFLOP/s; then one or two loops), the sweep stops and the direct solver
        # finishes from the current subspace.  Intervals inside the spectrum, where the Krylov sweeps stagnate, cost four
        # loops of them instead of fpm[4].
        plan = {}

        def abort_check(loop_idx, eps, elapsed):
            if loop_idx < 3:
                return False
            if "t_direct" not in plan:
                plan["t_direct"] = None
                if _band_direct_fits(eng, A, B, int(fpm[2])):
                    kl, ku, _nb, _blk = eng.band_plan()
                    if _blk == 2:      # multifrontal plan: measured 0.19 s for cfg 3 (1.05e11 flop per node, 16 nodes)
                        plan["t_direct"] = 0.08 + eng.direct_plan_flops() * int(fpm[2]) / 1.3e13
                    else:
                        plan["t_direct"] = 0.3 + (N / 128.0) * 1.5e-3 + 8.0 * N * kl * (kl + ku) * int(fpm[2]) / 2e13
            if plan["t_direct"] is None:
                return False
            fin = [e for e in eps if np.isfinite(e) and e > 0]
            tol = 10.0 ** (-int(fpm[3]))
            if len(fin) >= 3 and len(eps) >= 3 and all(np.isfinite(e) for e in eps[-3:]):
                rho = (eps[-1] / eps[-3]) ** 0.5
                remaining = math.inf if rho >= 1.0 else max(0.0, math.log(tol / eps[-1]) / math.log(rho)) * (elapsed / (loop_idx + 1))
            else:
                remaining = math.inf                                    # no Ritz value inside after four loops
            return remaining > 3.0 * plan["t_direct"]
    warm_start = bool(warm_start)                 # an explicitly named iterative solver keeps the reference's zero guess
    solver_maxiter = 500 if solver_maxiter is None else int(solver_maxiter)
    res = feast_hip_hermitian(eng, A, B, Emin, Emax, M0, fpm, solver=solver, solver_tol=solver_tol,
                              solver_maxiter=solver_maxiter, solver_restart=solver_restart,
                              warm_start=warm_start, inner_rtol=inner_rtol, real_projection=real_projection,
                              inner_precision=inner_precision, group=group, Q0=Q0, contour=contour,
                              contour_policy=contour_policy, eps_floor=float(np.sqrt(np.finfo(np.float32).eps)) if single else 0.0,
                              abort_check=abort_check)
    if (res.info == 5 and substituted is not None and substituted.get("used") in ("cocg", "bicgstab") and group is None
            and _band_direct_fits(eng, A, B, int(fpm[2]))):
        # The Krylov sweeps did not converge (typically an interval inside the spectrum: the shifted systems are then
        # indefinite and badly conditioned).  solver=:direct was what the caller asked for, and the direct solver for general
        # patterns fits the device: run it, as the reference's default would have from the start.
        krylov_info, krylov_loops = int(res.info), int(res.loop)
        # (the direct solver starts from the caller's / the seeded subspace, not from what the Krylov loops left: measured on
        #  cfg 3's interval around 2.0, the stagnated subspace carries a spurious pair into the direct solve -- M = 41, info 5
        #  after 20 loops -- where the fresh start converges in two)
        res = feast_hip_hermitian(eng, A, B, Emin, Emax, M0, fpm, solver="banded", solver_tol=solver_tol,
                                  real_projection=real_projection, inner_precision=inner_precision, group=group, Q0=Q0,
                                  contour=contour, eps_floor=float(np.sqrt(np.finfo(np.float32).eps)) if single else 0.0)
        substituted = dict(substituted, fallback=_direct_label(eng), krylov_info=krylov_info,
                           krylov_loops=krylov_loops)
    if substituted is not None and isinstance(res.stats, dict):
        res.stats["solver_substitution"] = substituted
    _release_band_factors(eng, keep_factors)
    if real_input:
        res = FeastResult(res.lambda_, np.real(res.q), res.M, res.res, res.info, res.epsout, res.loop, res.stats)
    if single:
        res = _demote(res, cplx_lambda=False)
    return res


def feast_general(A, B=None, center=0.0, radius=1.0, *, M0=10, fpm=None, backend="hip", solver="direct",
                  solver_tol=0.0, solver_maxiter=500, solver_restart=30, group=None, engine=None, device=0, Q0=None,
                  inner_precision=64, contour=None, keep_factors=False):
    """feast_general(A, [B,] center, radius; M0, fpm): src/interfaces/feast_interfaces.jl:274-379.
    ``keep_factors``: as in feast()."""
    if backend not in _BACKENDS:
        raise ValueError(f"Unknown backend '{backend}' (this package provides: hip)")
    if A.shape[0] != A.shape[1]:
        raise ValueError("Matrix A must be square")
    if not radius > 0:
        raise ValueError("radius must be positive")
    single = _single_precision(A, B)
    if single:
        A, B = _promote(A), _promote(B)
    fpm = feastinit() if fpm is None else fpm
    feastdefault(fpm)
    M0 = min(int(M0), A.shape[0])
    substituted = None
    if sp.issparse(A) and solver in ("direct", "lu"):
        # the reference factors z B - A with UMFPACK (src/sparse/feast_sparse.jl:943); here: banded LU for narrow
        # bands, else batched BiCGStab on the (non-symmetric) shifted systems with the reference's iterative
        # settings (zero guess, rtol = atol = 10^-fpm[3], src/sparse/feast_sparse.jl:164-203)
        solver = _sparse_direct_solver(A, B, int(fpm[8]))
        if solver == "dense":
            A, B, solver = _densify(A), _densify(B), "direct"
            substituted = {"requested": "direct", "used": "dense LU of the expanded matrix"}
        elif solver == "krylov":
            # unpreconditioned Krylov sweeps cannot solve shifted systems whose spectrum surrounds the shift (the usual case
            # for a contour inside a non-Hermitian spectrum): the direct solver for general patterns takes them whenever its
            # factors fit the device
            engine = _engine(engine, device)
            if _band_direct_fits(engine, A, B, int(fpm[8]), group, complexify=True):
                solver = "banded"
                substituted = {"requested": "direct", "used": _direct_label(engine)}
            else:
                solver = "bicgstab"
                substituted = {"requested": "direct", "used": solver, "warm_start": False, "inner_rtol": None,
                               "solver_maxiter": int(solver_maxiter)}
                _warn_substitution(substituted)
    elif solver == "krylov":
        solver = "bicgstab"
    elif solver == "sparse_direct":
        solver = "banded"
    eng = _engine(engine, device)
    res = feast_hip_general(eng, A, B, complex(center), float(radius), M0, fpm, solver=solver, inner_precision=inner_precision,
                             solver_tol=solver_tol, solver_maxiter=solver_maxiter,
                             solver_restart=solver_restart, group=group, Q0=Q0, contour=contour,
                             eps_floor=float(np.sqrt(np.finfo(np.float32).eps)) if single else 0.0)
    if substituted is not None and isinstance(res.stats, dict):
        res.stats["solver_substitution"] = substituted
    _release_band_factors(eng, keep_factors)
    if single:
        res = _demote(res, cplx_lambda=True)
    return res
