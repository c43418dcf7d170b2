"""The ``:hip`` backend: host-side FEAST refinement loops that keep the reference's state
machine and call the MI355X kernels through an *engine* (``engine.HipEngine``) for every
per-quadrature-node operation.

  feast_hip_hermitian  -- variant A ("QR + Rayleigh-Ritz"), mirrors
        _feast_dense_complex_hermitian  src/dense/feast_dense.jl:78-351
        _feast_sparse_hermitian         src/sparse/feast_sparse.jl:246-499
  feast_hip_general    -- variant C maths of feast_grci!/feast_gegv!
        src/kernel/feast_kernel.jl:646-962, src/dense/feast_dense.jl:402-593
  feast_hip_complex_symmetric -- complex-symmetric sibling of variant A (q^T instead of q^H)
        src/dense/feast_dense.jl:1026-1259, src/sparse/feast_sparse.jl:509-711

Quadrature nodes are block-partitioned over the ranks of the communicator attached to the engine
(``feasthip_comm_init_rank``) exactly like ``distribute_contour_points``
(src/parallel/feast_parallel.jl:433-447); each rank sweeps its nodes and the C ABI itself sums
Q_proj (and the per-node status) with ONE packed RCCL all-reduce over xGMI inside every
``contour_apply`` call -- the image of src/parallel/feast_mpi.jl:117-119 -- after which every rank
runs the reduced eigenproblem redundantly, as the MPI path does (src/parallel/feast_mpi.jl:121-139).
No ``torch.distributed`` collective is issued by this module; a ``group`` argument is only a
convenience to attach the engine's communicator through an existing process group.

The reduced M0 x M0 eigenproblem stays on host LAPACK.
"""
from __future__ import annotations

import math
import threading
import time

import numpy as np
import scipy.linalg as sla

from .contour import (balanced_contour_points, cost_balanced_contour_points, distribute_contour_points, feast_contour,
                      feast_gcontour, feast_inside_gcontour, split_balanced_assignment)
from .parameters import check_feast_srci_input, feast_tolerance, feastdefault
from .types import FeastError, FeastResult

SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)

try:
    # The reduced M0 x M0 problems are far too small for threaded BLAS, and a BLAS pool that keeps spinning
    # after the call starves the thread that feeds the GPU launch queue (measured on a 256-core host:
    # 0.89 s -> 0.71 s per cfg-3 solve).
    from threadpoolctl import ThreadpoolController as _ThreadpoolController
    _BLAS_POOLS = _ThreadpoolController()
except Exception:
    _BLAS_POOLS = None


class small_lapack:
    """Context manager: run the enclosed host LAPACK calls on one BLAS thread.  Re-entrant and cheap when nested: only the
    outermost level talks to threadpoolctl (setting and restoring the limits costs 0.1-0.3 ms, as much as the 64 x 64
    eigenproblem itself), so the drivers hold it for the whole solve and the per-loop calls nest inside it for free.
    The nesting depth is process wide (the BLAS limit is) and guarded by a lock: drivers running on several host threads
    (one engine each) share one limit, released when the last of them leaves."""
    _depth = 0
    _outer = None
    _lock = threading.Lock()

    def __enter__(self):
        cls = small_lapack
        with cls._lock:
            if cls._depth == 0 and _BLAS_POOLS is not None:
                cls._outer = _BLAS_POOLS.limit(limits=1)
                cls._outer.__enter__()
            cls._depth += 1
        return self

    def __exit__(self, *exc):
        cls = small_lapack
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0 and cls._outer is not None:
                ctx, cls._outer = cls._outer, None
                ctx.__exit__(None, None, None)
        return False


def seeded_subspace(N, M0, seed=20260515, complex_values=False):
    """Initial subspace: real Gaussian columns of unit norm (src/core/feast_tools.jl:6-43).
    The Julia MersenneTwister stream is not reproducible outside Julia; the structure is."""
    rng = np.random.default_rng([seed, N, M0, int(complex_values)])
    R = rng.standard_normal((N, M0))
    I = rng.standard_normal((N, M0)) if complex_values else None
    # (same random stream and the same values as the first version of this routine, in a third of the passes over the
    #  N x M0 block: the norms are taken on the real arrays and the result is assembled directly in column-major order --
    #  at N = 50 000, M0 = 64 this is a fifth of a default feast() call)
    if complex_values:
        nrm = np.sqrt(np.einsum("ij,ij->j", R, R) + np.einsum("ij,ij->j", I, I))
    else:
        nrm = np.sqrt(np.einsum("ij,ij->j", R, R))
    nrm[nrm == 0] = 1.0
    out = np.zeros((N, M0), dtype=np.complex128, order="F")
    out.real = R / nrm
    if complex_values:
        out.imag = I / nrm
    return out


def _world(engine, group=None):
    """(rank, world) of the communicator attached to ``engine``.  When none is attached but the host runs a
    ``torch.distributed`` group of more than one rank, the engine is attached through it first (control plane
    only: the unique id travels over the group, the reductions are the library's own)."""
    if getattr(engine, "comm_size", 1) > 1:
        return engine.comm_rank, engine.comm_size
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            engine.comm_init_from_group(group)
            return engine.comm_rank, engine.comm_size
    except ImportError:
        pass
    return 0, 1


def _reorder_by_interval(lam, Emin, Emax, n):
    """Stable inside-first permutation (src/core/feast_aux.jl:144-197) -> (perm, ninside)."""
    inside = [i for i in range(n) if Emin <= lam[i] <= Emax]
    outside = [i for i in range(n) if not (Emin <= lam[i] <= Emax)]
    return np.array(inside + outside, dtype=np.int64), len(inside)


try:
    from scipy.linalg.lapack import dsygvd as _dsygvd
except Exception:                                                # pragma: no cover
    _dsygvd = None


def _reduced_hermitian_eig(Sq, Aq):
    """eigen(Hermitian(Sq), Hermitian(Aq)) with the general fallback
    (src/dense/feast_dense.jl:270-284)."""
    with small_lapack():
        try:
            if not (np.any(Sq.imag) or np.any(Aq.imag)):
                # real-symmetric pencil (real projection of real-symmetric input): dsygvd called directly -- the same
                # eigenpairs as zhegv at a third of the time, without the argument checking of the scipy wrapper
                # (eigh: 0.26 ms for 64 x 64 on one BLAS thread, of which LAPACK itself is about half)
                a = np.array(Sq.real, dtype=np.float64, order="F")
                b = np.array(Aq.real, dtype=np.float64, order="F")
                if _dsygvd is not None:
                    lam, V, info = _dsygvd(a, b, itype=1, jobz="V", uplo="L", overwrite_a=1, overwrite_b=1)
                    if info == 0:
                        return np.asarray(lam, dtype=np.float64), V.astype(np.complex128)
                    raise np.linalg.LinAlgError("dsygvd info %d" % info)
                lam, V = sla.eigh(a, b)
                return np.asarray(lam, dtype=np.float64), V.astype(np.complex128)
            lam, V = sla.eigh(Sq, Aq)
            return np.asarray(lam, dtype=np.float64), V
        except Exception:
            w, V = sla.eig(Sq, Aq)
            return np.real(w).astype(np.float64), V


def feast_hip_hermitian(engine, A, B, Emin, Emax, M0, fpm, *, freeze_guards_after=None, reduced_solver="host",
                        solver="direct", solver_tol=0.0,
                        solver_maxiter=500, solver_restart=30, warm_start=True, inner_rtol=None,
                        real_projection=None, group=None, Q0=None, seed=20260515, contour=None, trace=None,
                        preloaded=False, node_assignment="block", inner_precision=64, column_groups=1,
                        spurious_filter=True, contour_policy=None, eps_floor=0.0, abort_check=None, resident_panels=True):
    """Variant A on the :hip engine.  Returns FeastResult (complex Ritz vectors, like
    _feast_dense_complex_hermitian; real-symmetric callers take real.(q) as the reference
    does, src/dense/feast_dense.jl:372-387).

    solver: "direct" (dense batched LU), "bicgstab"/"iterative" (batched BiCGStab), "gmres".
    solver_tol: 0 -> 10^-fpm[3] like the reference.  Krylov stop test: ||r|| <= tol + tol*||r0||.
    warm_start (iterative only, not in the reference): after the first loop the Ritz pairs
      (lambda_j, q_j) seed the solves with Y0 = q_j/(z_e - lambda_j), whose residual is
      r_j/(z_e - lambda_j) -- Galerkin-orthogonal to the current subspace.  ``inner_rtol``
      then bounds the reduction relative to that initial residual (default solver_tol).
    column_groups: g > 1 (iterative solvers, multi-rank) arranges the ranks as
      (world/g node groups) x (g column groups): a rank sweeps its node group for only its
      block of right-hand-side columns (blocks of >= 16 columns).  The reference shards nodes
      only (feast_parallel.jl:433-447); with Krylov solves the near-axis nodes need 10x the
      iterations of the others, so pure node sharding is bounded by the slowest node while
      columns of one node cost the same.  "auto" picks the largest g dividing the world size
      that leaves >= 16 columns per rank.  Q_proj columns are disjoint across column groups,
      so the one all-reduce per loop is unchanged.
    inner_precision: 64 | 32 (iterative solvers on sparse matrices).  32 solves the correction
      (z_e B - A) d = r0/||r0|| of each warm-started system in complex64 and adds it back in
      fp64; valid for inexact solves only (inner_rtol >= 1e-5).  Warm start, residuals,
      orthonormalisation and Rayleigh-Ritz stay fp64, so the converged eigenpairs are unchanged.
    real_projection: None -> True for real-symmetric A, B.  Q_proj = Re(sum 2 w_e Y_e), the
      full-contour FEAST filter (what the reference's real paths do, feast_parallel.jl:38-55,
      feast_kernel.jl:183-186).  False keeps variant A's complex half-contour sum
      (feast_dense.jl:231), whose filter only decays like 1/distance: same converged
      eigenpairs, many more refinement loops.
    contour_policy: None keeps fpm[18] as given (the reference's behaviour).  "auto" (inexact iterative solves with the
      real projection only; ignored otherwise) lets the driver pick the ellipse ratio fpm[18] itself, loop by loop
      (feasthip_policy_*, csrc/fh_policy.hpp): with inner solves that reduce the residual by inner_rtol per loop the contraction of a
      refinement loop is max(filter ratio, ~2 inner_rtol), so among the candidate ratios the one minimising the
      predicted work  a^-0.6 / ln(1 / max(filter ratio(a), inner_rtol))  is taken (a^-0.6: measured fall of the
      Krylov iterations per loop with the ratio a; a taller ellipse moves every node away from the spectrum).  The
      filter ratio is evaluated at the reach of the current subspace (feasthip_policy_reach of the Ritz values; loop 0:
      the a-priori 1.4 half widths of a subspace 1.5 times the eigenvalue count).  Safeguard: when a loop contracts
      the residual by less than 0.3 although the policy promised better, then -- if inner solves stopped at the iteration
      cap -- the cap is doubled, else the ratio is halved for the next loops, down to the reference's circle.
    """
    N = A.shape[0]
    feastdefault(fpm)
    info = check_feast_srci_input(N, M0, Emin, Emax)
    if info:
        return FeastResult(np.zeros(0), np.zeros((N, 0), dtype=np.complex128), 0, np.zeros(0), info, math.inf, 0)
    rank, world = _world(engine, group)
    iterative = solver not in ("direct", "lu", "banded")
    tol_value = feast_tolerance(fpm) if solver_tol == 0.0 else float(solver_tol)

    t_setup = time.perf_counter()
    if not preloaded:                            # matrices already resident on the device
        engine.set_problem(A, B)
    if contour is None:
        Zne, Wne = feast_contour(Emin, Emax, fpm)
    else:
        Zne, Wne = contour
    engine.set_contour(Zne, Wne, 2.0)            # weight = 2*Wne[e]: src/dense/feast_dense.jl:174
    if real_projection is None:
        import scipy.sparse as _sp
        _isc = lambda M_: M_ is not None and np.iscomplexobj(M_.data if _sp.issparse(M_) else M_)
        q_real = Q0 is None or (hasattr(Q0, "data_ptr") and not bool((Q0.imag != 0).any())) or \
            (not hasattr(Q0, "data_ptr") and (not np.iscomplexobj(Q0) or not np.any(np.imag(Q0))))
        real_projection = not (_isc(A) or _isc(B)) and q_real
    engine.set_real_projection(bool(real_projection))
    if column_groups == "auto":
        column_groups = 1
        if iterative and world > 1:
            for g in range(world, 0, -1):
                if world % g == 0 and M0 // g >= 16:
                    column_groups = g
                    break
    column_groups = int(column_groups)
    if column_groups < 1 or world % column_groups != 0 or (column_groups > 1 and not iterative):
        raise ValueError("column_groups must divide the world size and needs an iterative solver")
    node_groups = world // column_groups
    node_rank, col_rank = rank // column_groups, rank % column_groups
    if (node_assignment == "balanced" or callable(node_assignment)) and node_groups > 1:
        nodes_here = balanced_contour_points(len(Zne), node_groups)[node_rank]
        engine.set_node_list(nodes_here)
        count = len(nodes_here)
        local_nodes = list(nodes_here)
    else:
        first, count = distribute_contour_points(len(Zne), node_groups)[node_rank]
        engine.set_node_range(first, count)
        local_nodes = list(range(first, first + count))

    # this rank's column group (my_cg of my_cgs): fixed by the (node groups) x (column groups) grid, or re-derived every
    # loop by split_balanced_assignment, which splits only the heaviest nodes by columns
    my_cg, my_cgs = col_rank, column_groups
    split_layout = (callable(node_assignment) or node_assignment == "balanced") and column_groups == 1 and world > 1 and iterative
    node_parts = {}                                  # node -> column groups it was swept in (its iteration count arrives summed)

    def column_block(ncols):
        """[c0, c1) of this rank's column group: blocks in multiples of 16, remainder to the last."""
        if my_cgs == 1:
            return 0, ncols
        per = max(16, -(-ncols // my_cgs // 16) * 16) if ncols >= 16 * my_cgs else -(-ncols // my_cgs)
        c0 = min(ncols, my_cg * per)
        c1 = ncols if my_cg == my_cgs - 1 else min(ncols, c0 + per)
        return c0, c1
    engine.set_solver(solver, rtol=tol_value, atol=tol_value if iterative else 0.0, maxit=solver_maxiter,
                      restart=solver_restart, cache_factors=True)
    if inner_precision not in (32, 64):
        raise ValueError("inner_precision must be 32 or 64")
    inexact = bool(iterative and warm_start and inner_rtol is not None and float(inner_rtol) > 10.0 * tol_value)
    if iterative and warm_start:
        # inexact-solve mode: every loop reduces the (warm-started) residual by inner_rtol
        rt = tol_value if inner_rtol is None else float(inner_rtol)
        if inner_precision == 32 and rt < 1e-5:
            raise ValueError("inner_precision=32 needs inner_rtol >= 1e-5 (single-precision correction solves)")
        engine.set_solver(solver, rtol=rt, atol=0.0, maxit=solver_maxiter, restart=solver_restart,
                          factor_precision=inner_precision)
    elif inner_precision == 32:
        if solver in ("direct", "lu", "banded"):
            # dense LU / blocked band LU: complex64 factors + fp64 iterative refinement inside every solve
            engine.set_solver(solver, rtol=tol_value, atol=0.0, maxit=solver_maxiter, restart=solver_restart,
                              factor_precision=32, cache_factors=True)
        else:
            raise ValueError("inner_precision=32 needs a direct solver or the warm-started inexact iterative mode")
    # -- the host policy of the inexact mode (contour steering, iteration-cap guards, last-loop tolerance) lives under the C
    #    ABI: feasthip_policy_* (csrc/fh_policy.hpp).  Steering only where the filter is the real-projection filter.
    auto_contour = bool(contour_policy == "auto" and contour is None and inexact and real_projection and int(fpm[16]) in (0, 1))
    pol = None
    policy_hist, policy_reach = [], []
    if inexact:
        import ctypes as _C
        from . import _lib as _libmod
        _plib = _libmod.load_library()
        pol = _libmod.FeastHipPolicy()
        rcp = _plib.feasthip_policy_init(_C.byref(pol), float(Emin), float(Emax), int(fpm[2]), int(fpm[16]), float(inner_rtol),
                                         float(max(feast_tolerance(fpm), float(eps_floor))), int(solver_maxiter), int(auto_contour), int(fpm[18]))
        if rcp != 0:
            pol = None
    if auto_contour and pol is not None:
        fpm = fpm.copy()
        fpm[18] = int(pol.aspect)
        Zne, Wne = feast_contour(Emin, Emax, fpm)
        engine.set_contour(Zne, Wne, 2.0)
        engine.set_node_list(local_nodes)         # set_contour resets the node selection to "all"
        policy_hist.append(int(pol.aspect))
    t_setup = time.perf_counter() - t_setup

    if Q0 is not None and hasattr(Q0, "data_ptr"):
        dQ = Q0                                   # initial subspace already resident on the device (M0 x N); only ever read
    elif Q0 is None:
        # the seeded start block is a function of (N, M0, seed): generating it on the host takes 35 ms at N = 50 000, M0 = 64 --
        # a sixth of a default feast() call -- so an engine keeps the last one on the device for repeated calls
        key = (int(N), int(M0), int(seed))
        cache = getattr(engine, "_seed_cache", None)
        if cache is not None and cache[0] == key:
            dQ = cache[1]
        else:
            dQ = engine.upload(seeded_subspace(N, M0, seed))
            try:
                engine._seed_cache = (key, dQ)
            except AttributeError:
                pass
    else:
        dQ = engine.upload(np.asarray(Q0, dtype=np.complex128))
    maxloop = int(fpm[4])
    eps_tol = max(feast_tolerance(fpm), float(eps_floor))      # eps_floor: sqrt(eps(Float32)) for single-precision callers
    epsout, info, loop_count, M_found, active = math.inf, 0, 0, 0, M0
    lam_vec = np.zeros(M0)
    res_vec = np.zeros(M0)
    ritz_lambda = None
    inner_cap, loop_rtol = int(solver_maxiter), (float(inner_rtol) if inner_rtol is not None else None)
    dX = None
    stats = {"setup_seconds": t_setup, "krylov_iterations": 0, "spmm_calls": 0, "factorizations": 0,
             "solve_seconds": 0.0, "loops": [], "node_iterations": [], "node_lists": [], "local_nodes": [int(v) for v in local_nodes], "phase_seconds": {"apply": 0.0, "reduce": 0.0, "ortho": 0.0,
                                                                   "project": 0.0, "eig": 0.0, "ritz": 0.0}}
    ph = stats["phase_seconds"]
    tick = time.perf_counter

    epsout_mp = math.inf                     # outer residual of the previous loop (refinement tolerance of complex64 factors)
    t_loops = time.perf_counter()
    # The refinement loop with resident panels (engine.contour_apply_resident / rr_reduce_resident / rr_ritz_resident): one
    # 64-column panel, reduced eigenproblem on the host.  The per-primitive calls remain for wide subspaces (M0 > 64), the
    # device eigensolver and engines without the resident entry points.
    resident = bool(getattr(engine, "resident", False)) and M0 <= 64 and reduced_solver != "device" and resident_panels
    have_ritz = False
    # one BLAS thread for the whole solve; the `with` releases the process-wide limit on every way out, including an
    # exception from the engine inside the loop (FeastHipError, a poisoned handle)
    with small_lapack():
        for loop_idx in range(0, maxloop + 1):
            loop_count = loop_idx
            t_ = tick()
            lam_guess = ritz_lambda if (iterative and warm_start) else None
            col_mask = None
            if lam_guess is not None and freeze_guards_after is not None and loop_idx > freeze_guards_after:
                # guard columns (Ritz value outside the interval) keep their warm start q/(z - lambda): they
                # stay in the subspace, scaled by the filter value, but no solves are spent on them
                col_mask = np.array([1 if Emin <= lam_guess[c] <= Emax else 0 for c in range(active)], dtype=np.int32)
            if hasattr(engine, "set_column_mask"):
                engine.set_column_mask(col_mask)                    # one-shot: consumed by the sweep below
            if my_cgs > 1:
                c0, c1 = column_block(active)
                engine.set_column_block(c0, c1 - c0)
            if inner_precision == 32 and not iterative:
                # inexact FEAST on complex64 factors: the solves are refined only as far as the current outer residual needs
                # (no refinement in the first loop; the last loops reach the full tolerance)
                ref_tol = 1.0 if not math.isfinite(epsout_mp) else min(1.0, max(tol_value, 1e-2 * epsout_mp))
                engine.set_solver(solver, rtol=ref_tol, atol=0.0, maxit=solver_maxiter, restart=solver_restart,
                                  factor_precision=32, cache_factors=True)
            # one call = this rank's (nodes x column block) sweep + the packed all-reduce inside the C ABI:
            # dP and status come back summed over all ranks (status indexed by contour node when world > 1)
            if resident:
                # resident panels: Q_proj stays in the library in the kernels' layout; after the first loop the subspace is
                # the Ritz block the previous loop left there (dQ is None)
                dP = None
                status, st = engine.contour_apply_resident(dQ, active, lam_guess)
            else:
                dP, status, st = engine.contour_apply(dQ, active, lam_guess)
            if my_cgs > 1:
                engine.set_column_block(0, -1)
            ph["apply"] += tick() - t_
            stats["krylov_iterations"] += st.get("krylov_iterations", 0)
            stats["spmm_calls"] += st.get("spmm_calls", 0)
            stats["factorizations"] += st.get("factorizations", 0)
            stats["solve_seconds"] += st.get("seconds_solve", 0.0)
            if hasattr(engine, "last_node_iterations"):
                stats["node_iterations"].append([int(v) for v in engine.last_node_iterations(count)])
                stats["node_lists"].append([int(v) for v in local_nodes])
            local_fail = int(np.max(status)) if (world > 1 or count > 0) else 0
            if split_layout and loop_idx >= 1 and hasattr(engine, "last_global_node_iterations"):
                # node-only layout asked for: re-derive it from the measured iteration counts and let the heaviest nodes be
                # split by columns over several ranks when that lowers the largest share (contour.split_balanced_assignment);
                # the counts arrived in the tail of the packed all-reduce, summed over the ranks that swept a node
                costs = [float(v) / node_parts.get(e, 1) for e, v in enumerate(engine.last_global_node_iterations())]
                layout = node_assignment(costs, world) if callable(node_assignment) else split_balanced_assignment(costs, world, ncols=active)
                nodes_here, my_cg, my_cgs = layout[rank]
                node_parts = {e: k for nodes, _g, k in layout for e in nodes}
                stats["layout"] = [(list(map(int, nodes)), int(g), int(k)) for nodes, g, k in layout]
                if list(nodes_here) != list(local_nodes):
                    engine.set_node_list(nodes_here)
                    count, local_nodes = len(nodes_here), list(nodes_here)
                    stats["local_nodes"] = [int(v) for v in local_nodes]
            elif (node_assignment == "balanced" and node_groups > 1 and iterative and hasattr(engine, "last_global_node_iterations")
                    and loop_idx >= 1):
                # re-balance the node groups from the iteration counts the sweep just measured (they arrived in the tail of the
                # packed all-reduce and are identical on every rank): the slow near-axis nodes no longer share a group by accident
                nodes_here = cost_balanced_contour_points(engine.last_global_node_iterations(), node_groups)[node_rank]
                if list(nodes_here) != list(local_nodes):
                    engine.set_node_list(nodes_here)
                    count, local_nodes = len(nodes_here), list(nodes_here)
                    stats["local_nodes"] = [int(v) for v in local_nodes]
            if local_fail == 8 or (local_fail == 5 and not warm_start):
                # direct: singular shift -> info 8 (src/dense/feast_dense.jl:199-203);
                # reference GMRES failure -> info 5 (src/dense/feast_dense.jl:221-225)
                info = int(FeastError.Feast_ERROR_LAPACK if local_fail == 8 else FeastError.Feast_ERROR_NO_CONVERGENCE)
                break

            t_ = tick()
            if resident:
                # _feast_qr_compress! and the projections in one call: rank + (Q_o^H A Q_o, Q_o^H B Q_o)
                rank_q, Sq, Aq = engine.rr_reduce_resident(active, SQRT_EPS)
                ph["project"] += tick() - t_
            else:
                rank_q = engine.orthonormalize(dP, active, SQRT_EPS)       # _feast_qr_compress!
                ph["ortho"] += tick() - t_
            if rank_q == 0:
                info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                break
            if reduced_solver == "device" and rank_q <= 64 and trace is None and hasattr(engine, "rayleigh_ritz"):
                # project + reduced eigenproblem (Jacobi in LDS) + reorder + Ritz vectors + residuals in one call;
                # None: reduced B not positive definite -> the host path below (general fallback of the reference)
                t_ = tick()
                rr = engine.rayleigh_ritz(dP, rank_q, Emin, Emax, use_B=True)
                ph["ritz"] += tick() - t_
                if rr is not None:
                    dX, lam_sorted, M, res = rr
                    if M == 0 and not (iterative and warm_start):
                        info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                        break
                    lam_vec[:rank_q] = lam_sorted
                    if M > 0:
                        res_vec[:M] = res
                        epsout = float(res.max())
                    else:
                        epsout = math.inf
                    epsout_mp = epsout
                    M_found = M
                    stats["loops"].append({"loop": loop_idx, "rank": rank_q, "M": M, "epsout": epsout,
                                           "krylov_iterations": st.get("krylov_iterations", 0)})
                    if M > 0 and epsout <= eps_tol:
                        break
                    if loop_idx == maxloop:
                        info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                        break
                    active = rank_q
                    dQ = dX
                    ritz_lambda = lam_sorted.copy()
                    continue
            if not resident:
                t_ = tick()
                Sq, Aq = engine.project(dP, rank_q, bilinear=False, hermitize=True)
                ph["project"] += tick() - t_
            t_ = tick()
            try:
                lam_red, v_red = _reduced_hermitian_eig(Sq, Aq)
            except Exception:
                info = int(FeastError.Feast_ERROR_LAPACK)
                break
            ph["eig"] += tick() - t_
            perm, M = _reorder_by_interval(lam_red, Emin, Emax, rank_q)
            lam_sorted = lam_red[perm]
            V_sorted = np.asfortranarray(v_red[:, perm])
            if trace is not None:
                trace.append({"loop": loop_idx, "rank": rank_q, "M": M, "lambda": lam_sorted.copy(), "status": status.copy(),
                              "stats": dict(st),
                              "node_iterations": engine.last_node_iterations(count) if hasattr(engine, "last_node_iterations") else None,
                              "column_iterations": engine.last_column_iterations(count, active) if hasattr(engine, "last_column_iterations") else None})
            if M == 0 and not (iterative and warm_start):
                info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                break
            t_ = tick()
            if resident:
                dX, res = None, engine.rr_ritz_resident(rank_q, V_sorted, lam_sorted, M, normalize=True, use_B=True)
            else:
                dX, res = engine.ritz_residual(dP, rank_q, V_sorted, lam_sorted, M, normalize=True, use_B=True)
            n_spurious = 0
            if inexact and spurious_filter and loop_idx >= 1 and M > 1:
                # Inexact inner solves leave solver noise in the guard columns.  Its Ritz values are arbitrary; one that
                # lands inside the interval has an O(1) residual that never contracts and would hold epsout up forever
                # (variant A has no spurious-pair removal; with exact solves the guard columns are true eigen-directions
                # and stay outside).  A pair is set aside when its relative residual is > 0.1 AND > 100x the smallest
                # residual of the pairs inside: a true pair inside the interval sees a filter value >= 1/2 and contracts
                # with the others, it cannot sit at 10 % while another pair is 100x ahead.  Set-aside pairs stay in the
                # subspace and are re-examined every loop (measured on a random pencil: the ten true pairs contract by
                # ~1e-2 per loop while one to three noise pairs stay at residual 1).
                import ctypes as _C
                resc = np.ascontiguousarray(res, dtype=np.float64)
                flags_i = np.zeros(M, dtype=np.int32)
                n_spurious = int(_plib.feasthip_policy_set_aside(resc.ctypes.data_as(_C.c_void_p), int(M), flags_i.ctypes.data_as(_C.c_void_p)))
                flag = flags_i.astype(bool)
                if 0 < n_spurious < M:
                    order = np.concatenate([np.nonzero(~flag)[0], np.nonzero(flag)[0], np.arange(M, rank_q)])
                    lam_sorted = lam_sorted[order]
                    V_sorted = np.asfortranarray(V_sorted[:, order])
                    M = M - n_spurious
                    if resident:
                        res = engine.rr_ritz_resident(rank_q, V_sorted, lam_sorted, M, normalize=True, use_B=True)
                    else:
                        dX, res = engine.ritz_residual(dP, rank_q, V_sorted, lam_sorted, M, normalize=True, use_B=True)
                else:
                    n_spurious = 0
            ph["ritz"] += tick() - t_
            have_ritz = True
            lam_vec[:rank_q] = lam_sorted
            if M > 0:
                res_vec[:M] = res
                epsout = float(res.max())
            else:
                epsout = math.inf
            epsout_mp = epsout
            M_found = M
            stats["loops"].append({"loop": loop_idx, "rank": rank_q, "M": M, "epsout": epsout, "set_aside": n_spurious,
                                   "krylov_iterations": st.get("krylov_iterations", 0),
                                   "res_inside": np.array(res[:M], dtype=float).copy() if M > 0 else np.zeros(0)})
            if M > 0 and epsout <= eps_tol:
                break
            if loop_idx == maxloop:
                info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                break
            if abort_check is not None and world == 1 and abort_check(loop_idx, [l["epsout"] for l in stats["loops"]],
                                                                      time.perf_counter() - t_loops):
                # the caller has a cheaper way to finish (api.feast: the sparse direct solver): stop here
                info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                stats["aborted"] = True
                break
            if pol is not None:
                # one call decides the next sweep: iteration cap (stagnation guard; capped nodes under the contour policy), inner
                # tolerance (relaxed when the outer tolerance is within reach), and -- under the contour policy -- fpm[18]
                import ctypes as _C
                ritz_c = np.ascontiguousarray(lam_sorted[:rank_q], dtype=np.float64)
                prev_aspect, prev_cap, prev_rtol = int(pol.aspect), int(pol.inner_cap), float(pol.next_rtol)
                _plib.feasthip_policy_update(_C.byref(pol), float(epsout), int(M), int(int(np.max(status)) == 5),
                                             ritz_c.ctypes.data_as(_C.c_void_p), int(rank_q))
                if int(pol.inner_cap) != prev_cap or float(pol.next_rtol) != prev_rtol:
                    inner_cap, loop_rtol = int(pol.inner_cap), float(pol.next_rtol)
                    engine.set_solver(solver, rtol=loop_rtol, atol=0.0, maxit=inner_cap, restart=solver_restart,
                                      factor_precision=inner_precision)
                    if inner_cap != int(solver_maxiter):
                        stats["inner_cap"] = inner_cap
                if auto_contour:
                    if int(pol.aspect) != prev_aspect:
                        fpm[18] = int(pol.aspect)
                        Zne, Wne = feast_contour(Emin, Emax, fpm)
                        engine.set_contour(Zne, Wne, 2.0)
                        engine.set_node_list(local_nodes)
                    policy_hist.append(int(pol.aspect))
                    policy_reach.append(None if pol.last_reach < 0 else round(float(pol.last_reach), 3))
            active = rank_q
            dQ = dX                                   # Q_basis[:, 1:rank] = solutions[:, 1:rank]
            ritz_lambda = lam_sorted.copy()

    if auto_contour and pol is not None:
        stats["contour_policy"] = {"fpm18_per_loop": policy_hist, "cap": int(pol.cap), "reach": policy_reach}
    if hasattr(engine, "set_column_mask"):
        engine.set_column_mask(None)
    if M_found == 0 and info == 0:
        info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
    # only the M converged Ritz vectors cross PCIe (the block is column-major: the first M rows of the tensor)
    if resident and have_ritz and M_found > 0:
        q = engine.download(engine.export_resident(M_found), M_found)
    else:
        q = engine.download(dX[:M_found], M_found) if (dX is not None and M_found > 0) else np.zeros((N, 0), dtype=np.complex128)
    return FeastResult(lam_vec[:M_found].copy(), q, M_found, res_vec[:M_found].copy(), info, epsout, loop_count, stats)


def feast_hip_general(engine, A, B, Emid, r, M0, fpm, *, solver="direct", solver_tol=0.0, solver_maxiter=500,
                      solver_restart=30, group=None, Q0=None, seed=20260515, inner_precision=64, contour=None, eps_floor=0.0):
    """Variant C (general, full contour, no factor 2, no orthonormalisation, residual
    without B): src/kernel/feast_kernel.jl:752-950 driven as in src/dense/feast_dense.jl:468-584."""
    N = A.shape[0]
    feastdefault(fpm)
    if N <= 0:
        return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0), 1, math.inf, 0)
    if M0 <= 0 or M0 > N:
        return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0), 2, math.inf, 0)
    if not r > 0:
        return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0), 4, math.inf, 0)
    rank, world = _world(engine, group)
    iterative = solver not in ("direct", "lu", "banded")
    tol_value = feast_tolerance(fpm) if solver_tol == 0.0 else float(solver_tol)
    Ac = A.astype(np.complex128) if not np.iscomplexobj(A) else A
    Bc = None if B is None else (B.astype(np.complex128) if not np.iscomplexobj(B) else B)
    engine.set_problem(Ac, Bc)
    # caller-supplied nodes/weights: the reference's "x" drivers (feast_gcsrgvx!/feast_gegvx!, src/sparse/feast_sparse.jl:1008-)
    Zne, Wne = feast_gcontour(Emid, r, fpm) if contour is None else contour
    engine.set_contour(Zne, Wne, 1.0)
    engine.set_real_projection(False)
    first, count = distribute_contour_points(len(Zne), world)[rank]
    engine.set_node_range(first, count)
    if inner_precision == 32 and solver not in ("direct", "lu", "banded"):
        raise ValueError("inner_precision=32 (complex64 LU factors + fp64 refinement) needs a direct solver")
    engine.set_solver(solver, rtol=tol_value, atol=tol_value if iterative else 0.0, maxit=solver_maxiter,
                      restart=solver_restart, cache_factors=True, factor_precision=32 if inner_precision == 32 else 64)
    Q_host = seeded_subspace(N, M0, seed) if Q0 is None else np.asarray(Q0, dtype=np.complex128)
    dQ = engine.upload(Q_host)
    eps_tol = max(feast_tolerance(fpm), float(eps_floor))
    maxloop = int(fpm[4])
    loop = 0
    stats = {"krylov_iterations": 0, "factorizations": 0, "solve_seconds": 0.0}
    epsout = math.inf
    while True:
        if inner_precision == 32:
            # inexact FEAST: the complex64 solves are refined only as far as the current outer residual needs
            ref_tol = 1.0 if not math.isfinite(epsout) else min(1.0, max(1e-14, 1e-2 * epsout))
            engine.set_solver(solver, rtol=ref_tol, atol=0.0, maxit=solver_maxiter, restart=solver_restart,
                              cache_factors=True, factor_precision=32)
        dq, status, st = engine.contour_apply(dQ, M0, None)
        stats["krylov_iterations"] += st.get("krylov_iterations", 0)
        stats["factorizations"] += st.get("factorizations", 0)
        stats["solve_seconds"] += st.get("seconds_solve", 0.0)
        fail = int(np.max(status)) if (world > 1 or count > 0) else 0     # summed over the ranks inside the call
        if fail:
            return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0),
                               int(FeastError.Feast_ERROR_LAPACK if fail == 8 else FeastError.Feast_ERROR_NO_CONVERGENCE),
                               math.inf, loop, stats)
        Aq, Sq = engine.project(dq, M0, bilinear=False, hermitize=False)   # Aq = q^H A q, Sq = q^H B q
        try:
            with small_lapack():
                lam_red, v_red = sla.eig(Aq, Sq)                            # feast_kernel.jl:812
        except Exception:
            return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0),
                               int(FeastError.Feast_ERROR_LAPACK), math.inf, loop, stats)
        ins = [i for i in range(M0) if feast_inside_gcontour(lam_red[i], Emid, r, fpm)]
        M = len(ins)
        if M == 0:
            return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0),
                               int(FeastError.Feast_ERROR_NO_CONVERGENCE), math.inf, loop, stats)
        inset = set(ins)
        perm = np.array(ins + [i for i in range(M0) if i not in inset], dtype=np.int64)
        lam = lam_red[perm]
        V = np.asfortranarray(v_red[:, perm])
        # normalise ALL M0 columns (feast_kernel.jl:864-876); residual WITHOUT B (:899-906)
        dX, res = engine.ritz_residual(dq, M0, V, lam, M0, normalize=True, use_B=False)
        res = res[:M]
        epsout = float(res.max())
        if epsout <= eps_tol or loop >= maxloop:
            order = sorted(range(M), key=lambda i: abs(lam[i]) ** 2)         # feast_sort_general!
            X = engine.download(dX, M0)
            return FeastResult(lam[:M][order].copy(), X[:, :M][:, order].copy(), M, res[order].copy(), 0, epsout, loop, stats)
        loop += 1
        dQ = dX


def feast_hip_complex_symmetric(engine, A, B, Emid, r, M0, fpm, *, solver="direct", solver_tol=0.0,
                                solver_maxiter=500, solver_restart=30, group=None, Q0=None, seed=20260515):
    """Complex-symmetric sibling of variant A (A == A^T, B == B^T, complex): the loop of
    _feast_dense_complex_symmetric / its sparse twin (src/dense/feast_dense.jl:1026-1259,
    src/sparse/feast_sparse.jl:509-711).  Same kernels as the Hermitian path with the full
    contour (weights unscaled), pivoted-QR compression, the BILINEAR projection q^T A q, q^T B q
    (feasthip_project bilinear=1), general reduced eigenproblem on the host, inside-first
    reorder by the contour, all rank columns normalised, residual with B, sort by |lambda|^2.
    Sparse input may use solver="cocg": z B - A is complex symmetric for complex-symmetric A, B
    -- the COCG kernels need real A, B, so complex input goes through "bicgstab"/"gmres"."""
    import scipy.sparse as _sp
    N = A.shape[0]
    feastdefault(fpm)
    empty = lambda code, loop=0: FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0), code, math.inf, loop)
    if N <= 0:
        return empty(1)
    if M0 <= 0 or M0 > N:
        return empty(2)
    if not r > 0:
        return empty(4)
    for name, Mx in (("A", A), ("B", B)):
        if Mx is None:
            continue
        sym = (abs(Mx - Mx.T).max() == 0) if _sp.issparse(Mx) else np.array_equal(Mx, Mx.T)
        if not sym:                                          # check_complex_symmetric, feast_dense.jl:1038
            raise ValueError(f"Matrix {name} must be complex symmetric ({name} == transpose({name}))")
    rank_, world = _world(engine, group)
    iterative = solver not in ("direct", "lu", "banded")
    tol_value = feast_tolerance(fpm) if solver_tol == 0.0 else float(solver_tol)
    Ac = A.astype(np.complex128) if not np.iscomplexobj(A) else A
    Bc = None if B is None else (B.astype(np.complex128) if not np.iscomplexobj(B) else B)
    engine.set_problem(Ac, Bc)
    Zne, Wne = feast_gcontour(Emid, r, fpm)
    engine.set_contour(Zne, Wne, 1.0)
    engine.set_real_projection(False)
    first, count = distribute_contour_points(len(Zne), world)[rank_]
    engine.set_node_range(first, count)
    engine.set_solver(solver, rtol=tol_value, atol=tol_value if iterative else 0.0, maxit=solver_maxiter,
                      restart=solver_restart, cache_factors=True)
    dQ = engine.upload(seeded_subspace(N, M0, seed, complex_values=True) if Q0 is None else np.asarray(Q0, dtype=np.complex128))
    eps_tol = feast_tolerance(fpm)
    maxloop = int(fpm[4])
    info, epsout, M_found, active, loop_count = 0, math.inf, 0, M0, 0
    lam_vec = np.zeros(M0, dtype=np.complex128)
    res_vec = np.zeros(M0)
    dX = None
    stats = {"krylov_iterations": 0, "factorizations": 0, "solve_seconds": 0.0}
    for loop_idx in range(0, maxloop + 1):
        loop_count = loop_idx
        dP, status, st = engine.contour_apply(dQ, active, None)
        stats["krylov_iterations"] += st.get("krylov_iterations", 0)
        stats["factorizations"] += st.get("factorizations", 0)
        stats["solve_seconds"] += st.get("seconds_solve", 0.0)
        fail = int(np.max(status)) if (world > 1 or count > 0) else 0
        if fail:
            info = int(FeastError.Feast_ERROR_LAPACK if fail == 8 else FeastError.Feast_ERROR_NO_CONVERGENCE)
            break
        rank_q = engine.orthonormalize(dP, active, SQRT_EPS)                  # _feast_qr_compress!, :1163
        if rank_q == 0:
            info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
            break
        Ared, Bred = engine.project(dP, rank_q, bilinear=True, hermitize=False)   # q^T A q, q^T B q
        try:
            with small_lapack():
                lam_red, v_red = sla.eig(Ared, Bred)
        except Exception:
            info = int(FeastError.Feast_ERROR_LAPACK)
            break
        ins = [i for i in range(rank_q) if feast_inside_gcontour(lam_red[i], Emid, r, fpm)]
        M = len(ins)
        if M == 0:
            info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
            break
        inset = set(ins)
        perm = np.array(ins + [i for i in range(rank_q) if i not in inset], dtype=np.int64)
        lam_sorted = lam_red[perm]
        V = np.asfortranarray(v_red[:, perm])
        dX, res = engine.ritz_residual(dP, rank_q, V, lam_sorted, rank_q, normalize=True, use_B=True)
        lam_vec[:rank_q] = lam_sorted
        res_vec[:M] = res[:M]
        epsout = float(res[:M].max())
        M_found = M
        if epsout <= eps_tol:
            break
        if loop_idx == maxloop:
            info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
            break
        active = rank_q
        dQ = dX
    if M_found == 0 and info == 0:
        info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
    if dX is None or M_found == 0:
        return FeastResult(np.zeros(0, complex), np.zeros((N, 0), complex), 0, np.zeros(0), info, epsout, loop_count, stats)
    X = engine.download(dX, M_found)
    order = sorted(range(M_found), key=lambda i: abs(lam_vec[i]) ** 2)         # feast_sort_general!
    return FeastResult(lam_vec[:M_found][order].copy(), X[:, order].copy(), M_found, res_vec[:M_found][order].copy(),
                       info, epsout, loop_count, stats)


def pfeast_hip_moments(engine, A, B, Emin, Emax, M0, fpm, *, group=None, Q0=None, seed=20260515):
    """Variant B ("moments") on the :hip engine -- the loop of the reference's parallel drivers
    pfeast_sygv!/pfeast_scsrgv! (src/parallel/feast_parallel.jl:58-207, 450-572) with the
    per-node worker pfeast_solve_sparse_single_point (:717-751) replaced by one
    feasthip_contour_apply call that also returns the moment matrices
        Aq = Re sum_e 2 w_e Q^T Y_e,   Sq = Re sum_e 2 w_e z_e Q^T Y_e,   Q_proj = Re sum_e 2 w_e Y_e.
    Real-symmetric A, B only (as in the reference).  No orthonormalisation: the reduced pencil
    (Sq, Aq) is solved as is and X = Q_proj V; all M0 columns are carried to the next loop.
    The reference itself routes high-level dense calls away from this variant because it "does
    not currently match serial results" (src/core/feast_backend_utils.jl:115); it is mirrored for
    the seam, variant A (feast_hip_hermitian) is the robust path.
    """
    import scipy.sparse as _sp
    N = A.shape[0]
    feastdefault(fpm)
    info = check_feast_srci_input(N, M0, Emin, Emax)
    if info:
        return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), info, math.inf, 0)
    rank, world = _world(engine, group)
    Bm = B if B is not None else (_sp.identity(N, format="csr") if _sp.issparse(A) else np.eye(N))
    engine.set_problem(A, Bm)
    Zne, Wne = feast_contour(Emin, Emax, fpm)
    engine.set_contour(Zne, Wne, 2.0)
    engine.set_real_projection(True)                      # real.(...) of feast_parallel.jl:38-55
    first, count = distribute_contour_points(len(Zne), world)[rank]
    engine.set_node_range(first, count)
    engine.set_solver("direct" if not _sp.issparse(A) else "bicgstab", rtol=1e-13, atol=0.0, maxit=5000)
    work = np.real(seeded_subspace(N, M0, seed)) if Q0 is None else np.real(np.asarray(Q0))
    eps_tol = feast_tolerance(fpm)
    max_loops = int(fpm[4])
    lam = np.zeros(M0)
    res = np.zeros(M0)
    q = np.zeros((N, M0))
    for loop in range(1, max_loops + 1):
        dQ = engine.upload(work.astype(np.complex128))
        # Q_proj and both moment matrices come back summed over the ranks (feast_mpi.jl:117-119)
        dP, status, st, Aq, Sq = engine.contour_apply(dQ, M0, None, want_moments=True)
        Aq, Sq = np.real(Aq), np.real(Sq)
        Q_proj = np.real(engine.download(dP, M0))
        try:
            Su = np.triu(Sq) + np.triu(Sq, 1).T               # Symmetric(X) reads the upper triangle
            Au = np.triu(Aq) + np.triu(Aq, 1).T
            with small_lapack():
                lam_red, v_red = sla.eigh(Su, Au)
        except Exception:
            with small_lapack():
                w_, v_red = sla.eig(Sq, Aq)
            lam_red, v_red = np.real(w_), np.real(v_red)
        q = Q_proj @ v_red
        perm, M = _reorder_by_interval(lam_red, Emin, Emax, M0)
        lam = np.asarray(lam_red)[perm]
        q = q[:, perm]
        if M == 0:
            return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), int(FeastError.Feast_ERROR_NO_CONVERGENCE), 0.0, loop)
        for j in range(M):
            nrm = np.linalg.norm(q[:, j])
            if nrm > 0:
                q[:, j] /= nrm
        for j in range(M):                                    # feast_residual!, src/core/feast_tools.jl:726-755
            r_ = A @ q[:, j] - lam[j] * (Bm @ q[:, j])
            res[j] = np.linalg.norm(r_) / max(abs(lam[j]), 1.0)
        epsout = float(res[:M].max())
        if epsout <= eps_tol:
            order = np.argsort(lam[:M], kind="stable")        # feast_sort!
            return FeastResult(lam[:M][order].copy(), q[:, :M][:, order].copy(), M, res[:M][order].copy(), 0, epsout, loop)
        work = q[:, :M0].copy()
    M = int(sum(1 for i in range(M0) if Emin <= lam[i] <= Emax))
    return FeastResult(lam[:M].copy(), q[:, :M].copy(), M, res[:M].copy(), int(FeastError.Feast_ERROR_NO_CONVERGENCE),
                       float(res[:M].max()) if M else 0.0, max_loops)


def feast_hip_symmetric_kernel(engine, A, B, Emin, Emax, M0, fpm, *, solver="direct", solver_tol=0.0, solver_maxiter=500,
                               solver_restart=30, group=None, Q0=None, seed=20260515, contour=None):
    """What the real-symmetric RCI kernel ``feast_srci!`` (src/kernel/feast_kernel.jl:7-275) returns when its jobs are
    served -- the maths behind ``feast_sbgv!`` (src/banded/feast_banded.jl:87-175) -- as ONE device sweep per refinement
    loop instead of a FACTORIZE / SOLVE round trip per quadrature node: ``contour_apply(want_moments)`` yields
        Q_proj = Re sum_e 2 w_e Y_e,  Aq = Re sum_e 2 w_e Q^T Y_e,  Sq = Re sum_e 2 w_e z_e Q^T Y_e   (:146-169)
    summed over the ranks; then, as the kernel does: ``eigen(Sq, Aq)`` of the general reduced pencil (:175) with LAPACK's
    eigenvector scaling (the residuals below are taken on un-normalised Ritz vectors and depend on it), q = Q_proj Re(V)
    with no normalisation (:183-187), stable inside-first reorder (:189-215), residual ||A q - lambda q|| / max(|lambda|, 1)
    WITHOUT B (:244-252: generalized problems therefore run all fpm[4] loops and return info = 0), stop test with
    loop >= fpm[4] (:258), all M0 Ritz vectors carried to the next loop (:269), feast_sort! at the end."""
    import scipy.sparse as _sp
    N = A.shape[0]
    feastdefault(fpm)
    info = check_feast_srci_input(N, M0, Emin, Emax)
    if info:
        return FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), info, 0.0, 0)
    rank, world = _world(engine, group)
    engine.set_problem(A, B)
    Zne, Wne = feast_contour(Emin, Emax, fpm) if contour is None else contour
    engine.set_contour(Zne, Wne, 2.0)                     # weight = 2 * Wne[e], feast_kernel.jl:150
    engine.set_real_projection(True)                      # real(...) after the sweep, :166-169
    first, count = distribute_contour_points(len(Zne), world)[rank]
    engine.set_node_range(first, count)
    iterative = solver not in ("direct", "lu", "banded")
    tol_value = feast_tolerance(fpm) if solver_tol == 0.0 else float(solver_tol)
    engine.set_solver(solver, rtol=tol_value, atol=tol_value if iterative else 0.0, maxit=solver_maxiter,
                      restart=solver_restart, cache_factors=True)
    if Q0 is not None:                                    # fpm[5] = 1: the caller's columns, normalised (:68-80)
        work = np.array(np.real(Q0), dtype=np.float64)
        nrm = np.linalg.norm(work, axis=0)
        nrm[nrm == 0] = 1.0
        work = work / nrm
    else:
        work = np.real(seeded_subspace(N, M0, seed))
    eps_tol = feast_tolerance(fpm)
    maxloop = int(fpm[4])
    stats = {"krylov_iterations": 0, "factorizations": 0, "solve_seconds": 0.0}
    empty = lambda code, loop: FeastResult(np.zeros(0), np.zeros((N, 0)), 0, np.zeros(0), int(code), 0.0, loop, stats)
    loop = 0
    with small_lapack():
        while True:
            dQ = engine.upload(work)
            dP, status, st, Aq, Sq = engine.contour_apply(dQ, M0, None, want_moments=True)
            for k_, s_ in (("krylov_iterations", "krylov_iterations"), ("factorizations", "factorizations"), ("solve_seconds", "seconds_solve")):
                stats[k_] += st.get(s_, 0)
            if (int(np.max(status)) if (world > 1 or count > 0) else 0) != 0:
                return empty(FeastError.Feast_ERROR_LAPACK, loop)           # a failed SOLVE job, feast_banded.jl:137-147
            try:
                w, V = sla.eig(np.real(Sq), np.real(Aq))
            except Exception:
                return empty(FeastError.Feast_ERROR_LAPACK, loop)
            V = np.array(V, copy=True)
            for j in range(V.shape[1]):                                     # LAPACK ggev scaling: max |re| + |im| = 1
                s_ = np.max(np.abs(V[:, j].real) + np.abs(V[:, j].imag))
                if s_ > 0:
                    V[:, j] /= s_
            lam = np.real(w)
            perm, M = _reorder_by_interval(lam, Emin, Emax, M0)
            if M == 0:
                return empty(FeastError.Feast_ERROR_NO_CONVERGENCE, loop)
            lam = lam[perm]
            Vp = np.asfortranarray(np.real(V)[:, perm].astype(np.complex128))
            # q = Q_proj Re(V) for all M0 columns, residuals of the first M without B and without normalisation
            dX, res = engine.ritz_residual(dP, M0, Vp, lam, M, normalize=False, use_B=False)
            epsout = float(res.max())
            if epsout <= eps_tol or loop >= maxloop:
                X = np.real(engine.download(dX[:M], M))
                order = np.argsort(lam[:M], kind="stable")                  # feast_sort!
                return FeastResult(lam[:M][order].copy(), X[:, order].copy(), M, res[order].copy(), 0, epsout, loop, stats)
            loop += 1
            work = np.real(engine.download(dX, M0))


def pfeast_hip_hermitian_moments(engine, A, B, Emin, Emax, M0, fpm, *, solver="direct", solver_tol=0.0, solver_maxiter=500,
                                 solver_restart=30, group=None, Q0=None, seed=20260515):
    """Variant B for COMPLEX HERMITIAN input on the :hip engine -- the loop of the reference's MPI driver
    _mpi_feast_complex_hermitian! (src/parallel/feast_mpi.jl:796-909) with the per-rank worker
    mpi_compute_complex_hermitian_moments (:523-571) replaced by one ``contour_apply(want_moments)`` call:
        zAq = sum_e 2 w_e Q^H Y_e,  zSq = sum_e 2 w_e z_e Q^H Y_e,  Q_proj = sum_e 2 w_e Y_e   (complex, no real part),
    already summed over the ranks inside the C ABI (the three MPI.Allreduce of :856-858 are ONE packed reduce).
    Then, as the reference does: Hermitian parts of the moments, eigen(Hermitian(Sq), Hermitian(Aq)) with the general
    fallback, X = Q_proj V, inside-first reorder, the first M columns normalised, residuals with B, and ALL M0
    columns carried to the next loop (no orthonormalisation, no compression).  Any M0 (wider than 64 columns: the
    moment matrices are assembled block column by block column on the device).
    Sparse input with ``solver="direct"``: there is no sparse LU on the device, the systems go through the device
    GMRES with a purely relative 1e-13 stop.  This un-normalised iteration amplifies solver error (about 50x in the
    first loop, 3x per further loop: measured on the reference's fixture), so with Krylov solves the outer tolerance
    fpm[3] should not be asked below ~1e-10; dense input (batched LU) reaches 1e-12 like the reference."""
    import scipy.sparse as _sp
    N = A.shape[0]
    feastdefault(fpm)
    info = check_feast_srci_input(N, M0, Emin, Emax)
    if info:
        return FeastResult(np.zeros(0), np.zeros((N, 0), dtype=np.complex128), 0, np.zeros(0), info, math.inf, 0)
    rank, world = _world(engine, group)
    sparse = _sp.issparse(A)
    Ac = A.astype(np.complex128)
    Bc = B.astype(np.complex128) if B is not None else (_sp.identity(N, dtype=np.complex128, format="csr") if sparse else np.eye(N, dtype=np.complex128))
    engine.set_problem(Ac, Bc)
    Zne, Wne = feast_contour(Emin, Emax, fpm)
    engine.set_contour(Zne, Wne, 2.0)                     # weight = 2 * local_Wne[e], feast_mpi.jl:548
    engine.set_real_projection(False)
    first, count = distribute_contour_points(len(Zne), world)[rank]
    engine.set_node_range(first, count)
    iterative = solver not in ("direct", "lu", "banded")
    substituted = False
    if solver in ("direct", "lu") and sparse:
        # no sparse LU on the device: the reference's own iterative option, restarted GMRES
        # (solve_shifted_iterative!, feast_sparse.jl:164-203), device resident here
        solver, iterative, substituted = "gmres", True, True
    tol_value = feast_tolerance(fpm) if solver_tol == 0.0 else float(solver_tol)
    if substituted:
        # standing in for a DIRECT solve: purely relative stop at 1e-13 per column.  (The reference's iterative option
        # stops at atol + rtol*||b|| with atol = rtol = tol; columns of this un-normalised iteration shrink to 1e-8 and
        # an absolute 1e-12 then leaves them at 1e-4 relative -- measured: the outer residual grows 3x per loop.)
        engine.set_solver(solver, rtol=min(tol_value, 1e-13), atol=0.0, maxit=max(solver_maxiter, 2000), restart=solver_restart)
    else:
        engine.set_solver(solver, rtol=tol_value, atol=tol_value if iterative else 0.0, maxit=solver_maxiter,
                          restart=solver_restart, cache_factors=True)
    Q_basis = seeded_subspace(N, M0, seed, complex_values=True) if Q0 is None else np.asarray(Q0, dtype=np.complex128)
    dQ = engine.upload(Q_basis)
    eps_tol = feast_tolerance(fpm)
    lam_vec, res_vec = np.zeros(M0), np.zeros(M0)
    epsout, info, M_found, loop_count = math.inf, 0, 0, 0
    dX = None
    for loop_idx in range(0, int(fpm[4]) + 1):
        loop_count = loop_idx
        dP, status, st, zAq, zSq = engine.contour_apply(dQ, M0, None, want_moments=True)
        fail = int(np.max(status)) if (world > 1 or count > 0) else 0
        if fail:                                          # _mpi_success_count(...) != size, feast_mpi.jl:849-852
            info = int(FeastError.Feast_ERROR_LAPACK if not iterative else FeastError.Feast_ERROR_NO_CONVERGENCE)
            break
        Aq = 0.5 * (zAq + zAq.conj().T)                   # _feast_hermitian_part!
        Sq = 0.5 * (zSq + zSq.conj().T)
        try:
            lam_red, v_red = _reduced_hermitian_eig(Sq, Aq)
        except Exception:
            info = int(FeastError.Feast_ERROR_LAPACK)
            break
        perm, M = _reorder_by_interval(lam_red, Emin, Emax, M0)
        if M == 0:
            info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
            break
        lam_sorted = lam_red[perm]
        V_sorted = np.asfortranarray(np.asarray(v_red, dtype=np.complex128)[:, perm])
        # X = Q_proj V, the first M columns normalised, residual ||A x - lambda B x|| / max(|lambda|, 1) for them
        dX, res = engine.ritz_residual(dP, M0, V_sorted, lam_sorted, M, normalize=True, use_B=True)
        lam_vec[:] = lam_sorted
        res_vec[:M] = res
        epsout = float(res.max())
        M_found = M
        if epsout <= eps_tol:
            break
        if loop_idx == int(fpm[4]):
            info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
            break
        dQ = dX                                           # copyto!(Q_basis, solutions): all M0 columns
    if dX is None or M_found == 0:
        return FeastResult(np.zeros(0), np.zeros((N, 0), dtype=np.complex128), 0, np.zeros(0),
                           info or int(FeastError.Feast_ERROR_NO_CONVERGENCE), epsout, loop_count)
    X = engine.download(dX, M_found)
    order = np.argsort(lam_vec[:M_found], kind="stable")  # feast_sort!
    return FeastResult(lam_vec[:M_found][order].copy(), X[:, order].copy(), M_found, res_vec[:M_found][order].copy(),
                       info, epsout, loop_count)
