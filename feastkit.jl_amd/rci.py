"""Reverse-communication (RCI) surface of the ``:hip`` backend: the DEVICE side.

An RCI caller -- the reference's ``feast_srci!/hrci!/grci!`` state machines (src/kernel/feast_kernel.jl:7-962) and the
loops around them (src/banded/feast_banded.jl:87-175, src/dense/feast_dense.jl:468-584,
src/interfaces/feast_matfree.jl:203-254) -- stays on the Julia host.  What the backend supplies is the servicing of the
jobs such a caller issues:

* ``HipRciServer`` -- 10 FACTORIZE / 11 SOLVE -> ``feasthip_shifted_solve`` (dense LU or band LU cached per contour node,
  or the batched Krylov solvers for CSR input), 30 MULT_A / 40 MULT_B -> ``feasthip_matmul``;
* ``HipRciServer.linear_solver()`` -- the matrix-free callback ``(Y, z, X)`` of src/interfaces/feast_matfree.jl:149,225,697.

The Python mirrors of the job state machines and of their caller loops, which the parity tests need because there is no
Julia here, are test infrastructure: ``tests/rci_callers.py``.
"""
from __future__ import annotations

import numpy as np

from .types import FeastRCIJob

JOB_INIT = int(FeastRCIJob.Feast_RCI_INIT)
JOB_DONE = int(FeastRCIJob.Feast_RCI_DONE)
JOB_FACTORIZE = int(FeastRCIJob.Feast_RCI_FACTORIZE)
JOB_SOLVE = int(FeastRCIJob.Feast_RCI_SOLVE)
JOB_MULT_A = int(FeastRCIJob.Feast_RCI_MULT_A)
JOB_MULT_B = int(FeastRCIJob.Feast_RCI_MULT_B)


# ---------------------------------------------------------------------------------------------
# device job server
# ---------------------------------------------------------------------------------------------
class HipRciServer:
    """Services RCI jobs 10/11/30/40 on the MI355X for one (A, B) pair held by ``engine``.

    Host arrays in, host arrays out (the RCI contract keeps every array caller-owned); one H2D
    and one D2H of an N x m block per job.  Dense input: ``solver="direct"`` -- the LU of
    ``z B - A`` is cached per contour node after ``set_contour`` (job 10 is then free on every
    later refinement loop).  CSR input: ``solver`` in {"bicgstab", "cocg", "gmres"}.
    """

    def __init__(self, engine, A, B=None, *, solver="direct", rtol=1e-13, atol=0.0, maxit=5000, restart=30):
        self.engine = engine
        self.N = A.shape[0]
        self.b_identity = B is None
        engine.set_problem(A, B)
        engine.set_real_projection(False)
        engine.set_solver(solver, rtol=rtol, atol=atol, maxit=maxit, restart=restart)
        self.z = None
        self.solves = 0

    def set_contour(self, Zne, Wne, weight_scale=1.0):
        """Announce the shifts the RCI loop will visit: they get their own factor slots."""
        self.engine.set_contour(np.asarray(Zne), np.asarray(Wne), weight_scale)

    def factorize(self, Ze):                       # job 10
        self.z = complex(Ze)

    def solve(self, rhs, out, m, multiply_B=True):  # job 11
        """out[:, :m] = (z B - A)^-1 (B rhs[:, :m])   (multiply_B=False: rhs is used as is, the
        matrix-free contract of feast_matfree.jl:225)."""
        eng = self.engine
        dX = eng.upload(np.asarray(rhs)[:, :m])
        if multiply_B and not self.b_identity:
            dX = eng.matmul(1, dX, m)
        dY, rc = eng.shifted_solve(self.z, dX, m)
        self.solves += 1
        if rc != 0:
            return rc
        Y = eng.download(dY, m)
        out[:, :m] = Y if np.iscomplexobj(out) else Y.real
        return 0

    def mult(self, which, X, out, m):              # jobs 30 ("A") / 40 ("B")
        eng = self.engine
        dY = eng.matmul(0 if which == "A" else 1, eng.upload(np.asarray(X)[:, :m]), m)
        Y = eng.download(dY, m)
        out[:, :m] = Y if np.iscomplexobj(out) else Y.real

    def linear_solver(self):
        """The matrix-free callback ``linear_solver(Y, z, X)``: Y = (z B - A)^-1 X, all columns;
        raises on failure so the driver maps it to info = 8 (feast_matfree.jl:226-229)."""
        def solver(Y, z, X):
            self.factorize(z)
            rc = self.solve(X, Y, X.shape[1], multiply_B=False)
            if rc != 0:
                raise RuntimeError(f"feasthip shifted solve failed ({rc})")
        return solver
