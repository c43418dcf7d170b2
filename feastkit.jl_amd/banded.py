"""Banded drivers on the :hip backend (banded half): the entry points of
src/banded/feast_banded.jl with the LAPACK gbtrf!/gbtrs! pair replaced by the batched banded LU of
libfeasthip.so (``solver="direct"`` -> FEASTHIP_SOLVER_BANDED) and the band mat-vecs by the CSR
kernels.  Band storage is the reference's: upper (k+1) x N for symmetric / Hermitian matrices,
(2k+1) x N for general ones (feast_banded.jl:1-7, 205-271, 488-509).

  feast_sbgv / feast_sbev   real symmetric       -> the maths of feast_srci! (:9-186, 1410-1432)
  feast_hbgv / feast_hbev   complex Hermitian    -> variant A loop          (:385-403, 561-830)
  feast_gbgv / feast_gbev   general              -> full-contour loop       (:1548-1600, 1088-1385)
"""
from __future__ import annotations

import numpy as np

from .hip_backend import feast_hip_general, feast_hip_hermitian, feast_hip_symmetric_kernel
from .ingest import band_general_to_csr, band_upper_to_csr
from .parameters import feastdefault, feastinit


def _solver_keyword(solver):
    if solver == "direct":
        return "banded"
    if solver in ("gmres", "iterative"):
        return "gmres"
    raise ValueError(f"Unsupported solver '{solver}'. Use :direct, :gmres, or :iterative.")


def _engine(engine, device=0):
    if engine is not None:
        return engine
    from .engine import HipEngine
    return HipEngine(device)


def feast_sbgv(A, B, kla, klb, Emin, Emax, M0, fpm=None, *, solver="direct", solver_tol=0.0, solver_maxiter=500,
               solver_restart=30, engine=None):
    """Real symmetric banded generalized problem (feast_sbgv!, :9-186): the maths of the RCI kernel feast_srci! with the
    banded LU (one factorisation per contour node, cached) -- one device sweep per refinement loop instead of a job round
    trip per node (hip_backend.feast_hip_symmetric_kernel)."""
    fpm = feastinit() if fpm is None else fpm
    feastdefault(fpm)
    Ab, Bb = np.asarray(A, dtype=np.float64), None if B is None else np.asarray(B, dtype=np.float64)
    if Ab.shape[0] < kla + 1:
        raise ValueError("A matrix storage insufficient for kla")
    if Bb is not None and Bb.shape[0] < klb + 1:
        raise ValueError("B matrix storage insufficient for klb")
    Ac = band_upper_to_csr(Ab, kla)
    Bc = None if Bb is None else band_upper_to_csr(Bb, klb)
    return feast_hip_symmetric_kernel(_engine(engine), Ac, Bc, float(Emin), float(Emax), int(M0), fpm, solver=_solver_keyword(solver),
                                      solver_tol=solver_tol, solver_maxiter=solver_maxiter, solver_restart=solver_restart)


def feast_sbev(A, ka, Emin, Emax, M0, fpm=None, **kw):
    """Standard problem: B = I (feast_sbev!, :1410-1432)."""
    return feast_sbgv(A, None, ka, 0, Emin, Emax, M0, fpm, **kw)


def feast_hbgv(A, B, ka, kb, Emin, Emax, M0, fpm=None, *, solver="direct", solver_tol=0.0, solver_maxiter=500,
               solver_restart=30, engine=None, **kw):
    """Complex Hermitian banded problem: variant A (_feast_banded_complex_hermitian, :561-830)."""
    fpm = feastinit() if fpm is None else fpm
    Ac = band_upper_to_csr(np.asarray(A, dtype=np.complex128), ka, "hermitian")
    Bc = None if B is None else band_upper_to_csr(np.asarray(B, dtype=np.complex128), kb, "hermitian")
    return feast_hip_hermitian(_engine(engine), Ac, Bc, float(Emin), float(Emax), int(M0), fpm, solver=_solver_keyword(solver),
                               solver_tol=solver_tol, solver_maxiter=solver_maxiter, solver_restart=solver_restart, **kw)


def feast_hbev(A, ka, Emin, Emax, M0, fpm=None, **kw):
    return feast_hbgv(A, None, ka, 0, Emin, Emax, M0, fpm, **kw)


def feast_gbgv(A, B, ka, kb, Emid, r, M0, fpm=None, *, solver="direct", solver_tol=0.0, solver_maxiter=500,
               solver_restart=30, engine=None):
    """General banded problem on the full contour (feast_gbgv!, :1548-1559)."""
    fpm = feastinit() if fpm is None else fpm
    Ac = band_general_to_csr(np.asarray(A, dtype=np.complex128), ka)
    Bc = None if B is None else band_general_to_csr(np.asarray(B, dtype=np.complex128), kb)
    return feast_hip_general(_engine(engine), Ac, Bc, complex(Emid), float(r), int(M0), fpm, solver=_solver_keyword(solver),
                             solver_tol=solver_tol, solver_maxiter=solver_maxiter, solver_restart=solver_restart)


def feast_gbev(A, ka, Emid, r, M0, fpm=None, **kw):
    return feast_gbgv(A, None, ka, 0, Emid, r, M0, fpm, **kw)
