"""Contour nodes/weights on the host -- tiny, stays outside
the kernels exactly as in the reference (src/core/feast_tools.jl:212-371)."""
from __future__ import annotations

import math

import numpy as np


def _gauss(n):
    return np.polynomial.legendre.leggauss(n)   # FastGaussQuadrature.gausslegendre(n)


_ZOLOTAREV = None


def zolotarev_point(n, k):
    """Zolotarev node and weight k (1..n; k = 0: the constant term) on the unit disc --
    src/core/feast_tools.jl:182-210.  The constants are FEAST's libnum tables, shipped as data
    (zolotarev_tables.json, extracted by tests/golden/make_zolotarev_tables.py); for an n that
    is not tabulated the reference warns and falls back to a trapezoid-like rule (:196-209)."""
    global _ZOLOTAREV
    if _ZOLOTAREV is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "zolotarev_tables.json")) as f:
            _ZOLOTAREV = json.load(f)
    tab = _ZOLOTAREV.get(str(int(n)))
    if tab is not None:
        if k == 0:
            return 0j, complex(*tab["we0"])
        if 1 <= k <= len(tab["nodes"]):
            xr, xi, wr, wi = tab["nodes"][k - 1]
            return complex(xr, xi), complex(wr, wi)
    import warnings
    warnings.warn(f"Zolotarev quadrature not available for n={n}, using approximation")
    if k == 0:
        return 0j, 1 + 0j
    theta = math.pi * (2 * k - 1) / (2 * n)
    return complex(math.cos(theta), math.sin(theta)), complex(0.0, math.pi / n)


def feast_contour(Emin, Emax, fpm):
    """Half contour for Hermitian problems: src/core/feast_tools.jl:212-284."""
    ne, fpm16, fpm18 = int(fpm[2]), int(fpm[16]), int(fpm[18])
    r = (Emax - Emin) / 2.0
    Emid = Emin + r
    aspect = fpm18 * 0.01
    Zne = np.empty(ne, dtype=np.complex128)
    Wne = np.empty(ne, dtype=np.complex128)
    if fpm16 == 0:
        x, w = _gauss(ne)
    for e in range(ne):
        if fpm16 == 2:                       # Zolotarev: nodes/weights scaled by r (:263-266)
            zx, zw = zolotarev_point(ne, e + 1)
            Zne[e] = zx * r + Emid
            Wne[e] = zw * r
            continue
        if fpm16 == 0:
            theta = -math.pi / 2 * x[e] + math.pi / 2
            fac = 0.25 * w[e]
        else:
            theta = math.pi - (math.pi / ne) / 2 - (math.pi / ne) * e
            fac = 1.0 / (2 * ne)
        Zne[e] = Emid + r * math.cos(theta) + 1j * r * aspect * math.sin(theta)
        Wne[e] = fac * (r * 1j * math.sin(theta) + r * aspect * math.cos(theta))
    return Zne, Wne


def feast_gcontour(Emid, r, fpm):
    """Full contour for general problems: src/core/feast_tools.jl:286-371."""
    ne, fpm16, fpm18, fpm19 = int(fpm[8]), int(fpm[16]), int(fpm[18]), int(fpm[19])
    Emid = complex(Emid)
    aspect = fpm18 * 0.01
    rot = (fpm19 / 180.0) * math.pi
    nr = r * (math.cos(rot) + 1j * math.sin(rot))
    Zne = np.empty(ne, dtype=np.complex128)
    Wne = np.empty(ne, dtype=np.complex128)

    def point(theta, fac, e):
        Zne[e] = Emid + nr * math.cos(theta) + nr * 1j * aspect * math.sin(theta)
        Wne[e] = fac * (nr * 1j * math.sin(theta) + nr * aspect * math.cos(theta))

    if fpm16 == 0:
        nu = ne // 2
        if nu > 0:
            xu, wu = _gauss(nu)
        xl, wl = _gauss(ne - nu)
        for e in range(nu):
            point(-math.pi / 2 * xu[e] + math.pi / 2, 0.25 * wu[e], e)
        for e in range(nu, ne):
            i = e - nu
            point(math.pi / 2 * xl[i] - math.pi / 2, 0.25 * wl[i], e)
    else:
        for e in range(ne):
            point(math.pi - (2 * math.pi / ne) / 2 - (2 * math.pi / ne) * e, 1.0 / ne, e)
    return Zne, Wne


def feast_inside_gcontour(lam, Emid, r, fpm=None):
    """src/core/feast_tools.jl:623-650."""
    w = complex(lam) - complex(Emid)
    aspect, rot = 1.0, 0.0
    if fpm is not None and len(fpm) > 19:
        if fpm[18] > 0:
            aspect = fpm[18] * 0.01
        if fpm[19] != 0:
            rot = (fpm[19] / 180.0) * math.pi
    if rot != 0.0:
        w *= complex(math.cos(-rot), math.sin(-rot))
    x = w.real / r
    y = w.imag / (r * aspect)
    return x * x + y * y <= 1.0


def distribute_contour_points(ne, nw):
    """Contiguous block partition, first ne % nw parts get one extra:
    src/parallel/feast_parallel.jl:433-447 (returns (first, count) per worker, 0-based)."""
    per, rem = divmod(ne, nw)
    out, start = [], 0
    for i in range(nw):
        size = per + (1 if i < rem else 0)
        out.append((start, size))
        start += size
    return out


def balanced_contour_points(ne, nw):
    """Node lists that pair near-axis with far-axis nodes: worker i gets nodes i, 2nw-1-i,
    2nw+i, ... (a snake over the contour).  Krylov iteration counts grow towards the real
    axis, so contiguous blocks (distribute_contour_points) leave one worker with all the slow
    nodes.  Deviation from the reference partition; the reduced sum is unaffected."""
    out = [[] for _ in range(nw)]
    for e in range(ne):
        r, k = divmod(e, nw)
        out[k if r % 2 == 0 else nw - 1 - k].append(e)
    return out


def cost_balanced_contour_points(costs, nw):
    """Node lists from measured per-node costs (Krylov iterations of the previous sweep): longest-processing-time
    greedy -- nodes by decreasing cost (ties: lower index first) onto the currently lightest worker (ties: lower
    worker first).  Deterministic, so every rank derives the same lists from the same reduced counts.  Each worker
    keeps at least one node while there are enough nodes."""
    ne = len(costs)
    order = sorted(range(ne), key=lambda e: (-float(costs[e]), e))
    load = [0.0] * nw
    out = [[] for _ in range(nw)]
    for e in order:
        empty = [k for k in range(nw) if not out[k]]
        remaining = ne - sum(len(o) for o in out)
        if empty and remaining <= len(empty):
            k = empty[0]
        else:
            k = min(range(nw), key=lambda q: (load[q], q))
        out[k].append(e)
        load[k] += float(costs[e]) + 1.0          # +1: a node costs something even when it converged at once
    return [sorted(o) for o in out]


def split_balanced_assignment(costs, nw, ncols=64, min_cols=16, max_heavy=3):
    """Rank layout that may split the heaviest contour nodes by COLUMNS.  The reference shards whole nodes
    (feast_parallel.jl:433-447); with Krylov solves one near-axis node can cost more than a fair share of the whole
    sweep (cfg 3: node 15 takes 399 of 1 630 node-iterations, the fair share of 8 ranks is 204), and a node's columns
    are independent systems, so the `heavy` group -- the h heaviest nodes -- goes to k ranks that each take a block of
    ncols / k right-hand-side columns of every heavy node, and the other nodes are spread over the remaining ranks by
    longest-processing-time.  Returns ``[(nodes, column_group, column_groups)] * nw``; every (node, column) pair is
    owned by exactly one rank.  (h, k) is the pair with the smallest maximum load, k in {2, 4}, at least `min_cols`
    columns per rank; no split when it does not lower the maximum by 10 %.  Deterministic in `costs`."""
    ne = len(costs)
    c = [float(v) + 1.0 for v in costs]

    def lpt(nodes, workers):
        load = [0.0] * workers
        out = [[] for _ in range(workers)]
        for e in sorted(nodes, key=lambda e: (-c[e], e)):
            empty = [k for k in range(workers) if not out[k]]
            remaining = len(nodes) - sum(len(o) for o in out)
            k = empty[0] if (empty and remaining <= len(empty)) else min(range(workers), key=lambda q: (load[q], q))
            out[k].append(e)
            load[k] += c[e]
        return [sorted(o) for o in out], (max(load) if load else 0.0)

    base_lists, base_max = lpt(list(range(ne)), nw)
    best = (base_max, 0, 1)
    order = sorted(range(ne), key=lambda e: (-c[e], e))
    for k in (2, 4):
        if k >= nw or ncols // k < min_cols:
            continue
        for h in range(1, min(max_heavy, ne - (nw - k)) + 1):
            heavy = order[:h]
            rest = [e for e in range(ne) if e not in heavy]
            if len(rest) < nw - k:
                continue
            _, light_max = lpt(rest, nw - k)
            score = max(sum(c[e] for e in heavy) / k, light_max)
            if score < best[0] - 1e-12:
                best = (score, h, k)
    score, h, k = best
    if h == 0 or score > 0.9 * base_max:
        return [(nodes, 0, 1) for nodes in base_lists]
    heavy = sorted(order[:h])
    rest = [e for e in range(ne) if e not in heavy]
    light_lists, _ = lpt(rest, nw - k)
    return [(list(heavy), g, k) for g in range(k)] + [(nodes, 0, 1) for nodes in light_lists]


# ------------------------------------------------------------------------------------------------------------------
# Contour policy for INEXACT shifted solves: lives UNDER the C ABI (csrc/fh_policy.hpp, feasthip_policy_* in
# include/feasthip.h) so that every host shim calls the same code; these are thin wrappers for diagnostics and tests.
# (The numpy restatement the library is pinned against: tests/policy_reference.py.)
# ------------------------------------------------------------------------------------------------------------------
def filter_ratio(Emin, Emax, ne, fpm16, aspect, d_rel, inside=None):
    """Upper envelope of the rational filter beyond d_rel half widths from the midpoint, divided by its minimum over the
    wanted eigenvalues (feasthip_policy_filter_ratio)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load_library()
    ins = None if inside is None or len(inside) == 0 else np.ascontiguousarray(inside, dtype=np.float64)
    return float(lib.feasthip_policy_filter_ratio(float(Emin), float(Emax), int(ne), int(fpm16), int(aspect), float(d_rel),
                                                  ins.ctypes.data_as(C.c_void_p) if ins is not None else None, 0 if ins is None else len(ins)))


def subspace_reach(ritz, Emin, Emax, quantile=0.8):
    """Reach of the subspace beyond the interval in half widths (feasthip_policy_reach); None without guard Ritz values."""
    import ctypes as C
    from . import _lib
    lib = _lib.load_library()
    r = np.ascontiguousarray(ritz, dtype=np.float64)
    v = float(lib.feasthip_policy_reach(r.ctypes.data_as(C.c_void_p), len(r), float(Emin), float(Emax), float(quantile)))
    return None if v < 0 else v
