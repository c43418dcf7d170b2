"""TEST / BASELINE INFRASTRUCTURE (like everything under oracle/): the contour sweep of variant A with the quadrature
nodes farmed out to host processes -- the shape of the reference's `:threads` and `:distributed` backends
(src/parallel/feast_parallel.jl:586-630 `Threads.@threads for e in 1:ne`, :484-503 master sum of the per-node
contributions), used by bench.py's `cpu_baseline.all_cores` leg and by tests/test_oracle_golden.py.

One worker process per group of nodes (node e belongs to worker e mod P).  A worker keeps the sparse LU factors of its
nodes across refinement loops (the reference's serial path caches them, src/sparse/feast_sparse.jl:334-342; its threaded
path re-factors every loop -- the farm is the kinder of the two to the CPU), computes
    sum_{e in mine} 2 w_e (z_e B - A)^{-1} (B Q)
for the block Q the master publishes in shared memory, and writes its partial sum to its own shared-memory slot; the
master adds the slots in worker order (deterministic).  BLAS is pinned to one thread inside every worker: the
parallelism is over nodes, as in the reference.
"""
from __future__ import annotations

import multiprocessing as mp
import os
from multiprocessing import shared_memory

import numpy as np
import scipy.sparse as sp


def _worker(conn, A, B, nodes, Zne, Wne, N, M0, name_in, name_out):
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:                                  # pragma: no cover
        limiter = None
    import feast_oracle as fo
    shm_in = shared_memory.SharedMemory(name=name_in)
    shm_out = shared_memory.SharedMemory(name=name_out)
    buf_in = np.ndarray((N, M0), dtype=np.complex128, order="F", buffer=shm_in.buf)
    buf_out = np.ndarray((N, M0), dtype=np.complex128, order="F", buffer=shm_out.buf)
    factors = {}
    Ac = sp.csc_matrix(A, dtype=np.complex128)
    Bc = None if B is None else sp.csc_matrix(B, dtype=np.complex128)
    ident = sp.identity(N, dtype=np.complex128, format="csc") if Bc is None else None
    try:
        while True:
            msg = conn.recv()
            if msg[0] == "stop":
                break
            active = int(msg[1])
            basis = np.array(buf_in[:, :active])
            rhs = basis if Bc is None else Bc @ basis
            acc = np.zeros((N, active), dtype=np.complex128, order="F")
            nfac = 0
            ok = True
            for e in nodes:
                try:
                    if e not in factors:
                        factors[e] = fo._splu(Zne[e] * (ident if Bc is None else Bc) - Ac)
                        nfac += 1
                    Y = factors[e].solve(np.ascontiguousarray(rhs))
                    if not np.all(np.isfinite(Y)):
                        raise np.linalg.LinAlgError("singular shifted system")
                    acc += (2 * Wne[e]) * Y
                except Exception:
                    ok = False
                    break
            buf_out[:, :active] = acc
            conn.send(("done", ok, nfac))
    finally:
        shm_in.close()
        shm_out.close()
        if limiter is not None:
            limiter.unregister() if hasattr(limiter, "unregister") else None
        conn.close()


class NodeFarm:
    """farm = NodeFarm(A, B, Zne, Wne, M0, workers); Q_proj = farm.sweep(Q[:, :active]); farm.close()"""

    def __init__(self, A, B, Zne, Wne, M0, workers=None, timeout_s=900.0):
        self.timeout_s = float(timeout_s)
        self.N = A.shape[0]
        self.M0 = int(M0)
        ne = len(Zne)
        cores = os.cpu_count() or 1
        self.workers = max(1, min(ne, int(workers) if workers else cores))
        nbytes = self.N * self.M0 * 16
        self.shm_in = shared_memory.SharedMemory(create=True, size=nbytes)
        self.buf_in = np.ndarray((self.N, self.M0), dtype=np.complex128, order="F", buffer=self.shm_in.buf)
        self.shm_out, self.buf_out, self.procs, self.conns = [], [], [], []
        self.factorizations = 0
        ctx = mp.get_context("fork")                  # the matrices reach the workers by copy-on-write, not by pickling
        for w in range(self.workers):
            so = shared_memory.SharedMemory(create=True, size=nbytes)
            self.shm_out.append(so)
            self.buf_out.append(np.ndarray((self.N, self.M0), dtype=np.complex128, order="F", buffer=so.buf))
            parent, child = ctx.Pipe()
            nodes = list(range(w, ne, self.workers))
            p = ctx.Process(target=_worker, args=(child, A, B, nodes, np.array(Zne), np.array(Wne), self.N, self.M0,
                                                  self.shm_in.name, so.name), daemon=True)
            p.start()
            child.close()
            self.procs.append(p)
            self.conns.append(parent)

    def sweep(self, basis):
        active = basis.shape[1]
        self.buf_in[:, :active] = basis
        for c in self.conns:
            c.send(("sweep", active))
        ok = True
        for c in self.conns:
            if not c.poll(self.timeout_s):            # a wedged worker must not hang the caller
                raise TimeoutError("node worker did not answer within %.0f s" % self.timeout_s)
            tag, good, nfac = c.recv()
            ok = ok and good
            self.factorizations += nfac
        if not ok:
            raise np.linalg.LinAlgError("a node worker failed")
        out = np.zeros((self.N, active), dtype=np.complex128, order="F")
        for b in self.buf_out:                        # worker order: deterministic sum
            out += b[:, :active]
        return out

    def close(self):
        for c in self.conns:
            try:
                c.send(("stop",))
            except Exception:
                pass
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()                              # exactly the PID this object started
        for s in [self.shm_in] + self.shm_out:
            try:
                s.close()
                s.unlink()
            except Exception:
                pass
        self.procs, self.conns, self.shm_out = [], [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
