"""Test-side RCI CALLERS: Python mirrors of the reference's job state machines ``feast_srci!/hrci!/grci!``
(src/kernel/feast_kernel.jl:7-962; same job codes -- src/core/feast_types.jl:227-249 --, same in-place array contract, same
``fpm[50..53]`` scratch use; ``RciRefs`` carries what the Julia signature passes as ``Ref``s) and of the caller loops
around them (src/banded/feast_banded.jl:87-175, src/dense/feast_dense.jl:468-584, src/interfaces/feast_matfree.jl:203-254).

In a FeastKit.jl deployment this role stays on the Julia host; it lives under tests/ because the parity tests need a caller
for the device job server (``feastkit.jl_amd/rci.py::HipRciServer``) and there is no Julia in the image.  Not product code.

The reduced eigenproblems use LAPACK ``ggev`` through scipy, like Julia's ``eigen(A, B)``; scipy rescales generalized
eigenvectors to unit 2-norm, LAPACK (and therefore the reference) leaves the largest component at ``|re|+|im| = 1`` --
``_lapack_scaling`` restores that, because the srci/hrci residuals are taken on the un-normalised Ritz vectors and so depend
on the scaling.
"""
from __future__ import annotations

import zlib

import numpy as np
import scipy.linalg as sla

from feastkit_jl_amd.contour import feast_contour, feast_gcontour, feast_inside_gcontour
from feastkit_jl_amd.hip_backend import seeded_subspace, small_lapack
from feastkit_jl_amd.parameters import feast_tolerance, feastdefault
from feastkit_jl_amd.rci import (HipRciServer, JOB_DONE, JOB_FACTORIZE, JOB_INIT, JOB_MULT_A, JOB_MULT_B,   # noqa: F401
                                 JOB_SOLVE)
from feastkit_jl_amd.types import FeastError, FeastResult


class RciRefs:
    """The ``Ref`` arguments of the RCI kernels: ijob, Ze, epsout, loop, mode, info."""
    __slots__ = ("ijob", "Ze", "epsout", "loop", "mode", "info")

    def __init__(self):
        self.ijob = JOB_INIT
        self.Ze = 0j
        self.epsout = 0.0
        self.loop = 0
        self.mode = 0
        self.info = 0


class RciState:
    """FeastSRCIState / FeastHRCIState / FeastGRCIState (src/core/feast_types.jl): must be the
    same object across the calls of one solve."""

    def __init__(self):
        self.Zne = None
        self.Wne = None
        self.ne = 0
        self.e = 1
        self.M = 0
        self.initialized = False
        self.Q0 = None
        self.Q_proj = None
        self.zAq = None
        self.zSq = None
        self.eps = 0.0
        self.maxloop = 0
        self.mult_a_for_projection = False


def _unit_columns(X, M0, complex_values, tag):
    """fpm[5] == 1: normalise the user's columns, random fallback for zero columns
    (feast_kernel.jl:68-80, 448-460, 712-724)."""
    rng = np.random.default_rng([zlib.crc32(tag.encode()), X.shape[0], M0])
    for j in range(M0):
        nrm = np.linalg.norm(X[:, j])
        if nrm > 0:
            X[:, j] /= nrm
        else:
            v = rng.standard_normal(X.shape[0])
            if complex_values:
                v = v + 1j * rng.standard_normal(X.shape[0])
            X[:, j] = v / np.linalg.norm(v)


def _lapack_scaling(V):
    """Rescale eigenvector columns to LAPACK ggev's convention: max_i |re v_i| + |im v_i| = 1."""
    V = np.array(V, copy=True)
    for j in range(V.shape[1]):
        s = np.max(np.abs(V[:, j].real) + np.abs(V[:, j].imag))
        if s > 0:
            V[:, j] /= s
    return V


def _inside_first(flags):
    inside = [i for i, f in enumerate(flags) if f]
    outside = [i for i, f in enumerate(flags) if not f]
    return inside + outside, len(inside)


def feast_sort(lambda_, q, res, M, key=None):
    """feast_sort! / feast_sort_general! (src/core/feast_tools.jl:653-713): insertion sort of the
    first M eigenpairs, ascending in ``key`` (lambda, or |lambda|^2 for the general kernel).  An
    insertion sort is stable, so a stable argsort yields the same permutation."""
    if M <= 1:
        return
    k = np.asarray(lambda_[:M]) if key is None else key(np.asarray(lambda_[:M]))
    order = np.argsort(k, kind="stable")
    lambda_[:M] = np.asarray(lambda_[:M])[order]
    res[:M] = np.asarray(res[:M])[order]
    q[:, :M] = q[:, :M][:, order]


def _check_init(N, M0, ok_interval, bad_interval_code):
    if N <= 0:
        return int(FeastError.Feast_ERROR_N)
    if M0 <= 0 or M0 > N:
        return int(FeastError.Feast_ERROR_M0)
    if not ok_interval:
        return int(bad_interval_code)
    return 0


def _seed_subspace(N, M0, complex_values):
    Q = seeded_subspace(N, M0, complex_values=complex_values)
    return Q if complex_values else np.real(Q)


# ---------------------------------------------------------------------------------------------
# real symmetric: feast_srci!  (feast_kernel.jl:7-275)
# ---------------------------------------------------------------------------------------------
def feast_srci(refs, N, work, workc, Aq, Sq, fpm, Emin, Emax, M0, lambda_, q, res, state, contour=None):
    """One call of the real-symmetric RCI kernel.  Arrays are caller-owned and mutated in place:
    work (N x M0 real), workc (N x M0 complex), Aq/Sq (M0 x M0 real), lambda_ (M0), q (N x M0
    real), res (M0).  ``contour = (Zne, Wne)`` replaces the custom-contour registry of
    feast_srcix! (:277-289).

    Job protocol: on FACTORIZE the caller factors ``refs.Ze * B - A``; on SOLVE it overwrites
    ``workc[:, :M0]`` with ``(Ze B - A)^-1 (B work[:, :M0])``; on MULT_A it writes
    ``A q[:, :refs.mode]`` into ``work``.
    """
    ijob = refs.ijob
    if ijob == JOB_INIT:
        feastdefault(fpm)
        refs.info = _check_init(N, M0, Emin < Emax, FeastError.Feast_ERROR_EMIN_EMAX)
        if refs.info:
            return
        Zne, Wne = contour if contour is not None else feast_contour(Emin, Emax, fpm)
        state.Zne, state.Wne = np.array(Zne, dtype=np.complex128), np.array(Wne, dtype=np.complex128)
        state.ne, state.e, state.initialized = len(state.Zne), 1, True
        fpm[50], fpm[51], fpm[52], fpm[53] = 1, state.ne, 0, 1
        refs.loop = 0
        for arr in (Aq, Sq, lambda_, q, res, workc):
            arr[...] = 0
        if fpm[5] == 1:
            _unit_columns(work, M0, False, "fallback")
        else:
            work[:, :M0] = _seed_subspace(N, M0, False)
        state.Q0 = np.array(work[:, :M0], copy=True)
        state.Q_proj = np.zeros((N, M0), dtype=np.complex128)
        state.zAq = np.zeros((M0, M0), dtype=np.complex128)
        state.zSq = np.zeros((M0, M0), dtype=np.complex128)
        refs.Ze = complex(state.Zne[0])
        refs.ijob = JOB_FACTORIZE
        return

    if ijob == JOB_FACTORIZE:
        refs.ijob = JOB_SOLVE
        work[:, :state.Q0.shape[1]] = state.Q0
        return

    if ijob == JOB_SOLVE:
        e, ne = state.e, state.ne
        Mc = state.Q0.shape[1]
        if e == 1:
            state.Q_proj[...] = 0
            state.zAq[...] = 0
            state.zSq[...] = 0
        weight = 2 * state.Wne[e - 1]
        Y = workc[:, :Mc]
        state.Q_proj[:, :Mc] += weight * Y
        moment = state.Q0[:, :Mc].T @ Y                 # adjoint of a real block
        state.zAq[:Mc, :Mc] += weight * moment
        state.zSq[:Mc, :Mc] += state.Zne[e - 1] * (weight * moment)
        fpm[50] = e + 1
        state.e = e + 1
        if e < ne:
            refs.Ze = complex(state.Zne[e])
            refs.ijob = JOB_FACTORIZE
            return
        fpm[50] = 1
        state.e = 1
        Aq[:Mc, :Mc] = state.zAq[:Mc, :Mc].real
        Sq[:Mc, :Mc] = state.zSq[:Mc, :Mc].real
        try:
            with small_lapack():
                w, V = sla.eig(Sq[:Mc, :Mc], Aq[:Mc, :Mc])
            V = _lapack_scaling(V)
            lam = np.real(w)
            Aq[:Mc, :Mc] = np.real(V)
            q[:, :Mc] = state.Q_proj[:, :Mc].real @ Aq[:Mc, :Mc]
            perm, M = _inside_first([Emin <= lam[i] <= Emax for i in range(Mc)])
            lambda_[:Mc] = lam[perm]
            q[:, :Mc] = q[:, :Mc][:, perm]
            fpm[52] = M
            state.M = M
            if M == 0:
                refs.info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                refs.ijob = JOB_DONE
                fpm[53] = 0
                state.initialized = False
                return
            refs.ijob = JOB_MULT_A
            refs.mode = M
            return
        except (np.linalg.LinAlgError, ValueError):
            refs.info = int(FeastError.Feast_ERROR_LAPACK)
            refs.ijob = JOB_DONE
            fpm[53] = 0
            state.initialized = False
            return

    if ijob == JOB_MULT_A:
        M = int(fpm[52])
        for j in range(M):
            res[j] = np.linalg.norm(work[:, j] - lambda_[j] * q[:, j]) / max(abs(lambda_[j]), 1.0)
        refs.epsout = float(np.max(res[:M]))
        if refs.epsout <= feast_tolerance(fpm) or refs.loop >= fpm[4]:
            feast_sort(lambda_, q, res, M)
            refs.mode = M
            refs.ijob = JOB_DONE
            fpm[53] = 0
            state.initialized = False
            return
        refs.loop += 1
        Aq[...] = 0
        Sq[...] = 0
        work[:, :M0] = q[:, :M0]
        state.e = 1
        fpm[50] = 1
        state.Q0[...] = work[:, :M0]
        refs.Ze = complex(state.Zne[0])
        refs.ijob = JOB_FACTORIZE
        return

    if ijob == JOB_DONE:
        state.initialized = False
        return
    state.initialized = False
    raise ValueError(f"FEAST RCI kernel: invalid job code ijob={ijob}")


# ---------------------------------------------------------------------------------------------
# complex Hermitian: feast_hrci!  (feast_kernel.jl:397-644)
# ---------------------------------------------------------------------------------------------
def feast_hrci(refs, N, work, workc, zAq, zSq, fpm, Emin, Emax, M0, lambda_, q, res, state, contour=None):
    """Complex-Hermitian RCI kernel: the trial subspace lives in ``workc`` (complex), MULT_A
    writes ``A q[:, :mode]`` into ``workc``; zAq/zSq are complex M0 x M0."""
    ijob = refs.ijob
    if ijob == JOB_INIT:
        feastdefault(fpm)
        state.initialized = True
        refs.info = _check_init(N, M0, Emin < Emax, FeastError.Feast_ERROR_EMIN_EMAX)
        if refs.info:
            state.initialized = False
            return
        Zne, Wne = contour if contour is not None else feast_contour(Emin, Emax, fpm)
        state.Zne, state.Wne = np.array(Zne, dtype=np.complex128), np.array(Wne, dtype=np.complex128)
        state.ne, state.e, state.M = len(state.Zne), 1, 0
        state.eps, state.maxloop = feast_tolerance(fpm), int(fpm[4])
        refs.loop = 0
        for arr in (zAq, zSq, lambda_, q, res, work):
            arr[...] = 0
        if fpm[5] == 1:
            _unit_columns(workc, M0, True, "fallback_hrci")
        else:
            workc[:, :M0] = _seed_subspace(N, M0, True)
        state.Q0 = np.array(workc[:, :M0], copy=True)
        state.Q_proj = np.zeros((N, M0), dtype=np.complex128)
        refs.Ze = complex(state.Zne[0])
        refs.ijob = JOB_FACTORIZE
        return

    if ijob == JOB_FACTORIZE:
        refs.ijob = JOB_SOLVE
        workc[:, :state.Q0.shape[1]] = state.Q0
        return

    if ijob == JOB_SOLVE:
        e, ne = state.e, state.ne
        Mc = state.Q0.shape[1]
        if e == 1:
            state.Q_proj[...] = 0
        weight = 2 * state.Wne[e - 1]
        Y = workc[:, :Mc]
        state.Q_proj[:, :Mc] += weight * Y
        temp = state.Q0.conj().T @ Y
        zAq[:Mc, :Mc] += weight * temp
        zSq[:Mc, :Mc] += weight * state.Zne[e - 1] * temp
        state.e = e + 1
        if e < ne:
            refs.Ze = complex(state.Zne[e])
            refs.ijob = JOB_FACTORIZE
            return
        state.e = 1
        try:
            with small_lapack():
                w, V = sla.eig(zSq[:Mc, :Mc], zAq[:Mc, :Mc])
            V = _lapack_scaling(V)
            lam = np.real(w)
            q[:, :Mc] = state.Q_proj[:, :Mc] @ V
            perm, M = _inside_first([Emin <= lam[i] <= Emax for i in range(Mc)])
            lambda_[:Mc] = lam[perm]
            q[:, :Mc] = q[:, :Mc][:, perm]
            state.M = M
            if M == 0:
                refs.info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                refs.ijob = JOB_DONE
                state.initialized = False
                return
            refs.ijob = JOB_MULT_A
            refs.mode = M
            return
        except (np.linalg.LinAlgError, ValueError):
            refs.info = int(FeastError.Feast_ERROR_LAPACK)
            refs.ijob = JOB_DONE
            state.initialized = False
            return

    if ijob == JOB_MULT_A:
        M = state.M
        for j in range(M):
            res[j] = np.linalg.norm(workc[:, j] - lambda_[j] * q[:, j]) / max(abs(lambda_[j]), 1.0)
        refs.epsout = float(np.max(res[:M]))
        if refs.epsout <= state.eps or refs.loop >= state.maxloop:
            feast_sort(lambda_, q, res, M)
            refs.mode = M
            refs.ijob = JOB_DONE
            state.initialized = False
            return
        refs.loop += 1
        zAq[...] = 0
        zSq[...] = 0
        workc[:, :M0] = q[:, :M0]
        state.Q0[...] = q[:, :M0]
        refs.Ze = complex(state.Zne[0])
        refs.ijob = JOB_FACTORIZE
        return

    if ijob == JOB_DONE:
        state.initialized = False
        return
    state.initialized = False
    raise ValueError(f"FEAST RCI kernel (Hermitian): invalid job code ijob={ijob}")


# ---------------------------------------------------------------------------------------------
# general: feast_grci!  (feast_kernel.jl:646-962)
# ---------------------------------------------------------------------------------------------
def feast_grci(refs, N, work, workc, Aq, Sq, fpm, Emid, r, M0, lambda_, q, res, state, contour=None):
    """General (non-Hermitian) RCI kernel, full contour.  SOLVE: caller overwrites ``workc`` with
    ``(Ze B - A)^-1 (B workc)``; MULT_B / MULT_A: caller writes ``B q[:, :mode]`` /
    ``A q[:, :mode]`` into ``workc``.  Aq = Q^H A Q, Sq = Q^H B Q, reduced pencil (Aq, Sq)."""
    ijob = refs.ijob
    if ijob == JOB_INIT:
        feastdefault(fpm)
        refs.info = _check_init(N, M0, r > 0, FeastError.Feast_ERROR_EMID_R)
        if refs.info:
            return
        Zne, Wne = contour if contour is not None else feast_gcontour(Emid, r, fpm)
        state.Zne, state.Wne = np.array(Zne, dtype=np.complex128), np.array(Wne, dtype=np.complex128)
        state.contour_given = contour is not None
        fpm[50], fpm[51], fpm[52], fpm[53] = 1, len(state.Zne), 0, 1
        refs.loop = 0
        for arr in (Aq, Sq, lambda_, q, res):
            arr[...] = 0
        if fpm[5] == 1:
            _unit_columns(workc, M0, True, "fallback_grci")
        else:
            workc[:, :M0] = _seed_subspace(N, M0, True)
        work[...] = 0
        state.Q0 = np.array(workc[:, :M0], copy=True)
        state.initialized = True
        refs.Ze = complex(state.Zne[0])
        refs.ijob = JOB_FACTORIZE
        return

    if ijob == JOB_FACTORIZE:
        refs.ijob = JOB_SOLVE
        workc[:, :state.Q0.shape[1]] = state.Q0
        return

    if ijob == JOB_SOLVE:
        e, ne = int(fpm[50]), int(fpm[51])
        q[:, :M0] += state.Wne[e - 1] * workc[:, :M0]
        fpm[50] = e + 1
        if e < ne:
            refs.Ze = complex(state.Zne[e])
            refs.ijob = JOB_FACTORIZE
            return
        fpm[50] = 1
        work[...] = 0
        refs.ijob = JOB_MULT_B
        refs.mode = M0
        return

    if ijob == JOB_MULT_B:
        Sq[:M0, :M0] = q[:, :M0].conj().T @ workc[:, :M0]
        workc[...] = 0
        refs.ijob = JOB_MULT_A
        refs.mode = M0
        state.mult_a_for_projection = True
        return

    if ijob == JOB_MULT_A:
        if state.mult_a_for_projection:
            Aq[:M0, :M0] = q[:, :M0].conj().T @ workc[:, :M0]
            state.mult_a_for_projection = False
            try:
                with small_lapack():
                    w, V = sla.eig(Aq, Sq)
                V = _lapack_scaling(V)
                flags = [feast_inside_gcontour(w[i], Emid, r, fpm) for i in range(M0)]
                perm, M = _inside_first(flags)
                fpm[52] = M
                if M == 0:
                    refs.info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
                    refs.ijob = JOB_DONE
                    fpm[53] = 0
                    state.initialized = False
                    return
                X = q[:, :M0] @ V
                lambda_[:M0] = w[perm]
                X = X[:, perm]
                nrm = np.linalg.norm(X, axis=0)
                nrm[nrm == 0] = 1.0
                q[:, :M0] = X / nrm
                workc[...] = 0
                refs.ijob = JOB_MULT_A
                refs.mode = M
                return
            except (np.linalg.LinAlgError, ValueError):
                refs.info = int(FeastError.Feast_ERROR_LAPACK)
                refs.ijob = JOB_DONE
                fpm[53] = 0
                state.initialized = False
                return
        M = int(fpm[52])
        for j in range(M):
            res[j] = np.linalg.norm(workc[:, j] - lambda_[j] * q[:, j]) / max(abs(lambda_[j]), 1.0)
        refs.epsout = float(np.max(res[:M])) if M else 0.0
        if refs.epsout <= feast_tolerance(fpm) or refs.loop >= fpm[4]:
            feast_sort(lambda_, q, res, M, key=lambda x: np.abs(x) ** 2)
            refs.mode = M
            refs.ijob = JOB_DONE
            fpm[53] = 0
            state.initialized = False
            return
        refs.loop += 1
        state.Q0[...] = q[:, :M0]
        Aq[...] = 0
        Sq[...] = 0
        q[...] = 0
        workc[:, :M0] = state.Q0
        if not getattr(state, "contour_given", False):
            Zne, Wne = feast_gcontour(Emid, r, fpm)
            state.Zne, state.Wne = np.array(Zne, dtype=np.complex128), np.array(Wne, dtype=np.complex128)
        fpm[50] = 1
        refs.Ze = complex(state.Zne[0])
        refs.ijob = JOB_FACTORIZE
        return

    if ijob == JOB_DONE:
        state.initialized = False
        return
    state.initialized = False
    raise ValueError(f"FEAST RCI kernel (General): invalid job code ijob={ijob}")


# ---------------------------------------------------------------------------------------------
# caller loops over a job server (HipRciServer on the device, NumpyRciServer on the CPU)
# ---------------------------------------------------------------------------------------------
def _result(refs, lambda_, q, res):
    M = int(refs.mode)
    return FeastResult(np.array(lambda_[:M]), np.array(q[:, :M]), M, np.array(res[:M]), int(refs.info),
                       float(refs.epsout), int(refs.loop))


def rci_solve_symmetric(server, Emin, Emax, M0, fpm, *, contour=None, Q0=None, matrix_free=False):
    """Real-symmetric RCI caller loop with every job on the device (the loop of
    src/banded/feast_banded.jl:87-175; ``matrix_free=True``: that of feast_matfree_srci!,
    src/interfaces/feast_matfree.jl:203-254, whose SOLVE passes ``work`` unmultiplied)."""
    N = server.N
    refs, state = RciRefs(), RciState()
    work = np.zeros((N, M0), order="F")
    workc = np.zeros((N, M0), dtype=np.complex128, order="F")
    Aq, Sq = np.zeros((M0, M0)), np.zeros((M0, M0))
    lambda_, res = np.zeros(M0), np.zeros(M0)
    q = np.zeros((N, M0), order="F")
    if Q0 is not None:
        fpm[5] = 1
        work[:, :M0] = np.real(Q0)
    guard = 0
    while True:
        feast_srci(refs, N, work, workc, Aq, Sq, fpm, Emin, Emax, M0, lambda_, q, res, state, contour)
        guard += 1
        if refs.ijob == JOB_INIT or refs.ijob == JOB_DONE:
            break
        if guard == 1:
            server.set_contour(state.Zne, state.Wne, 2.0)
        if refs.ijob == JOB_FACTORIZE:
            server.factorize(refs.Ze)
        elif refs.ijob == JOB_SOLVE:
            rc = server.solve(work, workc, M0, multiply_B=not matrix_free)
            if rc != 0:
                refs.info = int(FeastError.Feast_ERROR_LAPACK)
                break
        elif refs.ijob == JOB_MULT_A:
            server.mult("A", q, work, refs.mode)
        else:
            raise ValueError(f"Unexpected FEAST RCI job code: ijob={refs.ijob}")
    return _result(refs, lambda_, q, res)


def rci_solve_hermitian(server, Emin, Emax, M0, fpm, *, contour=None, Q0=None):
    """Complex-Hermitian RCI caller loop (feast_hrci!) with every job on the device."""
    N = server.N
    refs, state = RciRefs(), RciState()
    work = np.zeros((N, M0), order="F")
    workc = np.zeros((N, M0), dtype=np.complex128, order="F")
    zAq, zSq = np.zeros((M0, M0), dtype=np.complex128), np.zeros((M0, M0), dtype=np.complex128)
    lambda_, res = np.zeros(M0), np.zeros(M0)
    q = np.zeros((N, M0), dtype=np.complex128, order="F")
    if Q0 is not None:
        fpm[5] = 1
        workc[:, :M0] = Q0
    guard = 0
    while True:
        feast_hrci(refs, N, work, workc, zAq, zSq, fpm, Emin, Emax, M0, lambda_, q, res, state, contour)
        guard += 1
        if refs.ijob == JOB_INIT or refs.ijob == JOB_DONE:
            break
        if guard == 1:
            server.set_contour(state.Zne, state.Wne, 2.0)
        if refs.ijob == JOB_FACTORIZE:
            server.factorize(refs.Ze)
        elif refs.ijob == JOB_SOLVE:
            rc = server.solve(workc, workc, M0)
            if rc != 0:
                refs.info = int(FeastError.Feast_ERROR_LAPACK)
                break
        elif refs.ijob == JOB_MULT_A:
            server.mult("A", q, workc, refs.mode)
        else:
            raise ValueError(f"Unexpected FEAST RCI job code: ijob={refs.ijob}")
    return _result(refs, lambda_, q, res)


def rci_solve_general(server, Emid, r, M0, fpm, *, contour=None, Q0=None):
    """General RCI caller loop -- feast_gegv!'s body (src/dense/feast_dense.jl:468-584) with the
    factorise / solve / multiply jobs on the device."""
    N = server.N
    refs, state = RciRefs(), RciState()
    work = np.zeros((N, M0), order="F")
    workc = np.zeros((N, M0), dtype=np.complex128, order="F")
    Aq, Sq = np.zeros((M0, M0), dtype=np.complex128), np.zeros((M0, M0), dtype=np.complex128)
    lambda_ = np.zeros(M0, dtype=np.complex128)
    res = np.zeros(M0)
    q = np.zeros((N, M0), dtype=np.complex128, order="F")
    if Q0 is not None:
        fpm[5] = 1
        workc[:, :M0] = Q0
    max_calls = None
    calls = 0
    while True:
        feast_grci(refs, N, work, workc, Aq, Sq, fpm, complex(Emid), float(r), M0, lambda_, q, res, state, contour)
        calls += 1
        if refs.ijob == JOB_INIT or refs.ijob == JOB_DONE:
            break
        if max_calls is None:
            server.set_contour(state.Zne, state.Wne, 1.0)
            max_calls = len(state.Zne) * (int(fpm[4]) + 1) * 10      # safety counter, feast_dense.jl:465
        if calls > max_calls:
            refs.info = int(FeastError.Feast_ERROR_NO_CONVERGENCE)
            break
        if refs.ijob == JOB_FACTORIZE:
            server.factorize(refs.Ze)
        elif refs.ijob == JOB_SOLVE:
            rc = server.solve(workc, workc, M0)
            if rc != 0:
                refs.info = int(FeastError.Feast_ERROR_LAPACK)
                break
        elif refs.ijob == JOB_MULT_B:
            server.mult("B", q, workc, refs.mode)
        elif refs.ijob == JOB_MULT_A:
            server.mult("A", q, workc, refs.mode)
        else:
            raise ValueError(f"Unexpected FEAST RCI job code: ijob={refs.ijob}")
    return _result(refs, lambda_, q, res)
