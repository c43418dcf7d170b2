"""TEST-ONLY job server for the RCI state machines: the method surface of
feastkit.jl_amd.rci.HipRciServer with exact dense numpy solves.  Lets the host-side job
protocol be checked on CPU against the oracle's straight-line restatement."""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp


class NumpyRciServer:
    def __init__(self, A, B=None):
        self.A = np.asarray(A.todense() if sp.issparse(A) else A, dtype=np.complex128)
        self.N = self.A.shape[0]
        self.B = np.eye(self.N, dtype=np.complex128) if B is None else np.asarray(B.todense() if sp.issparse(B) else B, dtype=np.complex128)
        self.factors = {}
        self.jobs = []
        self.z = None

    def set_contour(self, Zne, Wne, weight_scale=1.0):
        self.contour = (np.array(Zne), np.array(Wne), weight_scale)

    def factorize(self, Ze):
        self.jobs.append(10)
        self.z = complex(Ze)
        if self.z not in self.factors:
            self.factors[self.z] = sla.lu_factor(self.z * self.B - self.A)

    def solve(self, rhs, out, m, multiply_B=True):
        self.jobs.append(11)
        X = np.asarray(rhs)[:, :m].astype(np.complex128)
        if multiply_B:
            X = self.B @ X
        Y = sla.lu_solve(self.factors[self.z], X)
        out[:, :m] = Y if np.iscomplexobj(out) else Y.real
        return 0

    def mult(self, which, X, out, m):
        self.jobs.append(30 if which == "A" else 40)
        Y = (self.A if which == "A" else self.B) @ np.asarray(X)[:, :m]
        out[:, :m] = Y if np.iscomplexobj(out) else Y.real

    def linear_solver(self):
        def solver(Y, z, X):
            self.factorize(z)
            self.solve(X, Y, X.shape[1], multiply_B=False)
        return solver
