"""TEST-ONLY stand-in for feastkit.jl_amd.engine.HipEngine: the same method surface,
computed on the CPU with the oracle's arithmetic on torch CPU tensors.  It exists so the
host-side refinement loops, node partition and the all-reduce (gloo) can be tested without a
GPU.  It is never imported by the product package."""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import torch

import feast_oracle as fo


class OracleEngine:
    def __init__(self):
        self.last_stats = {}
        self.real_projection = False
        self.calls = {"contour_apply": 0}
        self.comm_size, self.comm_rank, self.group = 1, 0, None
        self.col_block = (0, -1)
        self.col_mask = None

    # the product engine owns its collective (RCCL inside the C ABI); this stand-in sums over a gloo group
    def comm_init_from_group(self, group=None, transport="auto"):
        import torch.distributed as dist
        self.group = group
        self.comm_rank, self.comm_size = dist.get_rank(group), dist.get_world_size(group)

    def set_column_block(self, first=0, count=-1):
        self.col_block = (int(first), int(count))

    def set_column_mask(self, mask):
        self.col_mask = None       # direct solves: nothing to skip

    def set_problem(self, A, B=None):
        self.sparse = sp.issparse(A)
        self.A = sp.csc_matrix(A, dtype=np.complex128) if self.sparse else np.asarray(A, dtype=np.complex128)
        self.B = None if B is None else (sp.csc_matrix(B, dtype=np.complex128) if self.sparse else np.asarray(B, dtype=np.complex128))
        self.N = A.shape[0]
        self.factors = {}

    def set_contour(self, Zne, Wne, scale):
        self.Zne, self.Wne, self.scale = np.array(Zne), np.array(Wne), scale
        self.ne = len(Zne)
        self.first, self.count = 0, self.ne
        self.node_list = list(range(self.ne))
        self.factors = {}

    def set_real_projection(self, on):
        self.real_projection = bool(on)

    def set_node_range(self, first, count):
        self.first, self.count = first, count
        self.node_list = list(range(first, first + count))

    def set_node_list(self, indices):
        self.node_list = [int(i) for i in indices]
        self.first, self.count = (self.node_list[0] if self.node_list else 0), len(self.node_list)

    def set_solver(self, solver="direct", **kw):
        self.solver = solver

    def free_factors(self):
        self.factors = {}
        self.calls["free_factors"] = self.calls.get("free_factors", 0) + 1

    def empty(self, m):
        return torch.zeros((m, self.N), dtype=torch.complex128)

    def upload(self, Q):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(Q, dtype=np.complex128).T))

    def download(self, dQ, m=None):
        a = dQ.numpy()
        if m is not None:
            a = a[:m]
        return np.asfortranarray(a.T)

    def _solve(self, e, rhs):
        z = self.Zne[e]
        if e not in self.factors:
            if self.sparse:
                Bm = sp.identity(self.N, dtype=np.complex128, format="csc") if self.B is None else self.B
                self.factors[e] = spla.splu(sp.csc_matrix(z * Bm - self.A))
            else:
                Bm = np.eye(self.N) if self.B is None else self.B
                self.factors[e] = sla.lu_factor(z * Bm - self.A)
        return self.factors[e].solve(np.ascontiguousarray(rhs)) if self.sparse else sla.lu_solve(self.factors[e], rhs)

    def contour_apply(self, dQ, m, ritz_lambda=None, want_moments=False):
        self.calls["contour_apply"] += 1
        Q = self.download(dQ)[:, :m]
        c0, cnt = self.col_block
        c1 = m if cnt < 0 else min(m, c0 + cnt)
        c0 = min(c0, m)
        rhs = Q if self.B is None else self.B @ Q
        P = np.zeros((self.N, m), dtype=np.complex128)
        nodes = getattr(self, 'node_list', range(self.first, self.first + self.count))
        if c1 > c0:
            for e in nodes:
                P[:, c0:c1] += self.scale * self.Wne[e] * self._solve(e, rhs[:, c0:c1])
        if self.real_projection:
            P = P.real.astype(np.complex128)
        out = torch.zeros((dQ.shape[0], self.N), dtype=torch.complex128)
        out[:m] = torch.from_numpy(np.ascontiguousarray(P.T))
        status = np.zeros(max(1, self.ne), dtype=np.int32)
        self.last_stats = {"krylov_iterations": 0, "spmm_calls": 0, "factorizations": 0, "seconds_solve": 0.0}
        zA = zS = None
        if want_moments:
            zA = np.zeros((m, m), dtype=np.complex128); zS = np.zeros((m, m), dtype=np.complex128)
            for e in nodes:
                G = Q.conj().T @ self._solve(e, rhs)
                zA += self.scale * self.Wne[e] * G
                zS += self.scale * self.Wne[e] * self.Zne[e] * G
            if self.real_projection:
                zA, zS = zA.real.astype(np.complex128), zS.real.astype(np.complex128)
        if self.comm_size > 1:
            # the ONE packed reduce of the C ABI: [Q_proj | zAq | zSq | status flags]
            import torch.distributed as dist
            parts = [torch.view_as_real(out).reshape(-1)]
            if want_moments:
                parts += [torch.from_numpy(np.ascontiguousarray(zA)).view(torch.float64).reshape(-1),
                          torch.from_numpy(np.ascontiguousarray(zS)).view(torch.float64).reshape(-1)]
            parts.append(torch.from_numpy(status.astype(np.float64)))
            pack = torch.cat(parts)
            dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=self.group)
            n0 = parts[0].numel()
            out = torch.view_as_complex(pack[:n0].reshape(dQ.shape[0], self.N, 2).contiguous())
            off = n0
            if want_moments:
                k = 2 * m * m
                zA = pack[off:off + k].numpy().view(np.complex128).reshape(m, m).copy(); off += k
                zS = pack[off:off + k].numpy().view(np.complex128).reshape(m, m).copy(); off += k
            status = (pack[off:].numpy() > 0).astype(np.int32) * 5
        if want_moments:
            return out, status, self.last_stats, zA, zS
        return out, status, self.last_stats

    def orthonormalize(self, dQ, m, rank_tol):
        Q = self.download(dQ)[:, :m]
        basis, rank = fo.qr_compress(Q, m, rank_tol)
        dQ[:rank] = torch.from_numpy(np.ascontiguousarray(basis.T))
        return rank

    def project(self, dQ, r, bilinear=False, hermitize=True):
        q = self.download(dQ)[:, :r]
        qt = q.T if bilinear else q.conj().T
        Aq = qt @ (self.A @ q)
        if self.B is None:
            Bq = np.eye(r, dtype=np.complex128) if (hermitize and not bilinear) else qt @ q
        else:
            Bq = qt @ (self.B @ q)
        if hermitize and not bilinear:
            Aq, Bq = fo.hermitian_part(Aq), fo.hermitian_part(Bq)
        return np.asfortranarray(Aq), np.asfortranarray(Bq)

    def ritz_residual(self, dQ, r, V, lam, M, normalize=True, use_B=True):
        q = self.download(dQ)[:, :r]
        X = q @ np.asarray(V)
        if normalize:
            for j in range(M):
                n = np.linalg.norm(X[:, j])
                if n > 0:
                    X[:, j] /= n
        res = fo.feast_residual(self.A, self.B if use_B else None, np.asarray(lam), X, M)
        out = torch.zeros((dQ.shape[0], self.N), dtype=torch.complex128)
        out[:r] = torch.from_numpy(np.ascontiguousarray(X.T))
        return out, res
