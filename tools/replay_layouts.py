"""What would one rank's share of cfg 3 cost on its own GPU?  Replays the contour sweeps of a converged one-GPU solve.

The driver launches N > 1 ranks only on an 8-GPU node that is not ours.  What one GPU can measure exactly is the time ONE
rank's share of every sweep takes when it has the GPU to itself -- which is the situation of every rank in the real run.
This tool runs the bench's cfg-3 solve once (world = 1) while recording the input of every contour sweep (the subspace
panel, the Ritz values of the warm start) and the per-node iteration counts, then replays every sweep for every rank of
a layout -- (node list, column block) through feasthip_set_node_list / feasthip_set_column_block -- and times it.  A
layout's estimated step is  sum over loops of [max over ranks of the share's sweep time] + the measured Rayleigh-Ritz
time per loop + a ring all-reduce estimate for the packed buffer.

Layouts (per world size W):
  grid g     (W / g node groups) x (g column groups), node groups re-balanced from the measured costs (what bench.py's
             column_groups="auto" picks for g = largest divisor of W with >= 16 columns per rank)
  nodes      node-only longest-processing-time lists from the measured costs (column_groups = 1)
  split      contour.split_balanced_assignment: node-only, the heaviest nodes split by columns

Usage: python tools/replay_layouts.py [W ...]      (default 2 4 8)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import feastkit_jl_amd as fk
from feastkit_jl_amd.contour import (balanced_contour_points, cost_balanced_contour_points, split_balanced_assignment)
import bench

worlds = [int(v) for v in sys.argv[1:]] or [2, 4, 8]
NE, M0 = bench.NE, bench.M0
torch.cuda.set_device(0)
eng = fk.HipEngine(0)
A, B, lam_exact = bench.build_problem()
eng.set_problem(A, B)
Q0_dev = eng.upload(fk.seeded_subspace(A.shape[0], M0))

record = []
_apply = eng.contour_apply


def recording_apply(dQ, m, ritz_lambda=None, want_moments=False):
    record.append({"dQ": dQ.clone(), "m": int(m), "lam": None if ritz_lambda is None else np.array(ritz_lambda, copy=True)})
    out = _apply(dQ, m, ritz_lambda, want_moments)
    record[-1]["its"] = [int(v) for v in eng.last_node_iterations(NE)]
    return out


def solve(resident):
    fpm = fk.feastinit()
    fpm[2], fpm[4], fpm[16], fpm[18] = NE, 40, 0, 4000
    return fk.feast_hip_hermitian(eng, A, B, bench.EMIN, bench.EMAX, M0, fpm, solver="cocg", warm_start=True, inner_rtol=3e-2,
                                  solver_maxiter=50, preloaded=True, Q0=Q0_dev, real_projection=True, resident_panels=resident)


solve(True)                              # warm-up: workspaces allocated
# the sweeps are recorded through the per-primitive loop (the resident loop does not pass its panels through contour_apply;
# the sweeps themselves are the same kernels), the Rayleigh-Ritz time per loop is the resident loop's
eng.contour_apply = recording_apply
eng.synchronize(); t0 = time.perf_counter()
res = solve(False)
eng.synchronize(); t_step = time.perf_counter() - t0
eng.contour_apply = _apply
res_r = solve(True)
ph = res_r.stats["phase_seconds"]
rr_per_loop = (ph["ortho"] + ph["project"] + ph["eig"] + ph["ritz"]) / max(1, int(res_r.loop) + 1)
print(f"one GPU: {t_step * 1e3:.1f} ms per step incl. recording copies, {len(record)} sweeps, M = {res.M}, info = {res.info}, "
      f"Rayleigh-Ritz {rr_per_loop * 1e3:.2f} ms per loop", flush=True)


def column_block(ncols, g, k):
    if k == 1:
        return 0, ncols
    per = max(16, -(-ncols // k // 16) * 16) if ncols >= 16 * k else -(-ncols // k)
    c0 = min(ncols, g * per)
    return c0, (ncols if g == k - 1 else min(ncols, c0 + per))


def time_share(rec, nodes, g, k):
    eng.set_node_list(list(nodes))
    c0, c1 = column_block(rec["m"], g, k)
    best = 1e9
    for _ in range(2):
        if k > 1:
            eng.set_column_block(c0, c1 - c0)
        eng.synchronize(); t = time.perf_counter()
        _apply(rec["dQ"], rec["m"], rec["lam"])
        eng.synchronize(); best = min(best, time.perf_counter() - t)
        if k > 1:
            eng.set_column_block(0, -1)
    return best


def layouts_for(W):
    out = {}
    for g in (1, 2, 4):
        if W % g or M0 // g < 16:
            continue
        ng = W // g

        def grid(costs, g=g, ng=ng):
            lists = balanced_contour_points(NE, ng) if costs is None else cost_balanced_contour_points(costs, ng)
            return [(lists[r // g], r % g, g) for r in range(W)]
        out["nodes" if g == 1 else f"grid {ng}x{g}"] = grid

    def split(costs):
        if costs is None:
            return [(nodes, 0, 1) for nodes in balanced_contour_points(NE, W)]
        return split_balanced_assignment(costs, W, ncols=M0)
    out["split"] = split
    return out


packed_bytes = A.shape[0] * M0 * 8               # real projection: one double per entry
for W in worlds:
    ring = 2.0 * (W - 1) / W * packed_bytes / 60e9 + 30e-6      # ring all-reduce over xGMI at ~60 GB/s effective per link pair
    print(f"== world {W}  (all-reduce estimate {ring * 1e3:.2f} ms per loop)", flush=True)
    for name, fn in layouts_for(W).items():
        total, worst = 0.0, []
        for li, rec in enumerate(record):
            costs = None if li < 2 else record[li - 1]["its"]       # the driver re-balances after loop 1 from that loop's counts
            layout = fn(costs)
            seen = {}
            times = []
            for nodes, g, k in layout:
                key = (tuple(nodes), g, k)
                if key not in seen:
                    seen[key] = time_share(rec, nodes, g, k)
                times.append(seen[key])
            total += max(times)
            worst.append(int(np.argmax(times)))
        est = total + len(record) * (rr_per_loop + ring)
        print(f"   {name:10s} sweeps {total * 1e3:7.1f} ms  -> step ~{est * 1e3:6.1f} ms = {res.M / est:6.0f} eigenpairs/s   "
              f"(last layout {fn(record[-2]['its'])})", flush=True)
eng.set_node_list(list(range(NE)))
