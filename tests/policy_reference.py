"""TEST-ONLY numpy restatement of the inexact-mode contour policy (the product's copy lives under the C ABI:
feastkit.jl_amd/csrc/fh_policy.hpp, feasthip_policy_*).  tests/test_host_logic.py pins the library against these
functions: filter values, envelope ratio, subspace reach, and whole policy trajectories."""
import math

import numpy as np

from feastkit_jl_amd.contour import feast_contour


# Contour policy for INEXACT shifted solves (not in the reference; used when `solver=:direct` on large sparse input
# maps to the batched Krylov solvers, api.py).  With exact solves the contraction of FEAST's subspace iteration per
# refinement loop is the filter ratio rho(lambda_{M0+1}) / rho(lambda_inside) -- 1e-5 and better for a 16-point Gauss
# rule on the circle.  With inner solves that only reduce the residual by `inner_rtol` per loop the contraction is
# max(filter ratio, inner_rtol): a filter sharper than inner_rtol is paid for (its nodes sit next to the real axis,
# where the shifted systems are worst conditioned) and never used.  The policy therefore picks the TALLEST ellipse
# (fpm[18], the reference's own parameter: src/core/feast_parameters.jl:232-247, src/core/feast_tools.jl:212-284)
# whose filter still separates the subspace from the rest of the spectrum by the factor the inner solves deliver.
ASPECT_CANDIDATES = (100, 150, 200, 300, 400, 600, 800, 1200, 1600, 2400, 3200, 4000, 5000, 6000, 8000)


def filter_values(Zne, Wne, lam):
    """rho(lambda) = Re sum_e 2 w_e / (z_e - lambda): the rational filter of the half contour with the real projection
    (src/parallel/feast_parallel.jl:38-55 take the real part; weight 2 w_e: src/dense/feast_dense.jl:174)."""
    lam = np.atleast_1d(np.asarray(lam, dtype=np.float64))
    return np.real((2.0 * np.asarray(Wne)[None, :] / (np.asarray(Zne)[None, :] - lam[:, None])).sum(axis=1))


_FILTER_TABLES = {}


def _filter_table(ne, fpm16, aspect):
    """(Zne, Wne) of the unit interval (-1, 1) and the outer envelope E(d) = max_{d' >= d} |rho(d')| on a log-spaced
    grid 1 <= d <= 60 (d in half widths from the midpoint).  The filter is invariant under shift and scaling of the
    interval and symmetric about its midpoint for these contours, so one table per (ne, fpm16, aspect) serves every
    solve; it is computed once per process."""
    key = (int(ne), int(fpm16), int(aspect))
    tab = _FILTER_TABLES.get(key)
    if tab is None:
        fpm = np.zeros(65, dtype=np.int64)
        fpm[2], fpm[16], fpm[18] = ne, fpm16, aspect
        Zne, Wne = feast_contour(-1.0, 1.0, fpm)
        d = np.exp(np.linspace(0.0, math.log(60.0), 4000))
        env = np.maximum.accumulate(np.abs(filter_values(Zne, Wne, d))[::-1])[::-1]
        tab = _FILTER_TABLES[key] = (Zne, Wne, d, env)
    return tab


def filter_ratio(Emin, Emax, ne, fpm16, aspect, d_rel, inside=None):
    """Upper envelope of |rho| over |lambda - Emid| >= d_rel * r, divided by the smallest |rho| over the wanted
    eigenvalues (`inside`: their current Ritz values; None: the whole interval, whose ends carry rho = 1/2)."""
    Zne, Wne, d, env = _filter_table(ne, fpm16, aspect)
    r = 0.5 * (Emax - Emin)
    mid = Emin + r
    out = env[min(len(d) - 1, int(np.searchsorted(d, max(d_rel, 1.0), side="left")))]
    pts = np.linspace(-1.0, 1.0, 65) if inside is None or len(inside) == 0 else (np.asarray(inside, dtype=np.float64) - mid) / r
    inn = np.abs(filter_values(Zne, Wne, pts)).min()
    return float(out / max(inn, 1e-300))


def choose_aspect(Emin, Emax, ne, fpm16, d_rel, target, inside=None, candidates=ASPECT_CANDIDATES):
    """The largest fpm[18] among `candidates` whose filter_ratio is <= target; 100 (the reference's circle) when none
    qualifies.  d_rel: where the first eigenvalue OUTSIDE the subspace is believed to lie, as a multiple of the
    interval's half width measured from its midpoint."""
    best = 100
    for a in candidates:
        if a >= 100 and filter_ratio(Emin, Emax, ne, fpm16, a, d_rel, inside) <= target:
            best = max(best, a)
    return int(best)


def subspace_reach(ritz, Emin, Emax, quantile=0.8):
    """How far the current subspace reaches beyond the interval, as a multiple of its half width measured from the
    midpoint: the `quantile` point of the distances of the guard Ritz values (those outside [Emin, Emax]).  The
    subspace holds the M0 eigen-directions with the largest filter values and the filter is symmetric about the
    midpoint, so the first eigenvalue outside the subspace lies beyond the outermost guard on EITHER side; the
    outermost guards are the least converged Ritz values (contaminated by far eigenvectors, they overshoot outward),
    hence a quantile instead of the maximum.  None when there are no guards."""
    ritz = np.asarray(ritz, dtype=np.float64)
    r = 0.5 * (Emax - Emin)
    mid = Emin + r
    g = np.sort(np.abs(ritz[(ritz < Emin) | (ritz > Emax)] - mid))
    if len(g) == 0:
        return None
    return float(g[min(len(g) - 1, int(quantile * len(g)))] / r)


def policy_pick(Emin, Emax, ne, fpm16, inner_rtol, cap, d_rel, inside=None, limit=None):
    """fpm[18] minimising the predicted work a^-0.6 / ln(1 / max(filter ratio, inner_rtol)) among the candidates."""
    best, best_cost = 100, None
    for a in ASPECT_CANDIDATES:
        if a > cap or (limit is not None and a > limit):
            continue
        c = max(filter_ratio(Emin, Emax, ne, fpm16, a, d_rel, inside), inner_rtol)
        if c >= 0.5:
            continue
        cost = a ** -0.6 / math.log(1.0 / c)
        if best_cost is None or cost < best_cost:
            best, best_cost = a, cost
    return int(best)


class PolicyReference:
    """The per-solve state machine of feasthip_policy_init / feasthip_policy_update, restated."""

    def __init__(self, Emin, Emax, ne, fpm16, inner_rtol, outer_tol, maxiter, steer, fpm18):
        self.Emin, self.Emax, self.ne, self.q, self.rt, self.tol = Emin, Emax, ne, fpm16, inner_rtol, outer_tol
        self.steer = bool(steer) and fpm16 in (0, 1)
        self.cap, self.inner_cap, self.base = 8000, maxiter, maxiter
        self.hist, self.prev, self.next_rtol = [], math.inf, inner_rtol
        self.aspect = policy_pick(Emin, Emax, ne, fpm16, inner_rtol, self.cap, 1.4) if self.steer else fpm18

    def update(self, eps, M, capped, ritz):
        self.hist.append(eps)
        self.hist = self.hist[-3:]
        if len(self.hist) >= 3 and self.hist[2] > 0.5 * self.hist[0] and self.inner_cap < 16 * self.base:
            self.inner_cap *= 2
            self.hist = []
        if self.steer:
            if math.isfinite(self.prev) and math.isfinite(eps) and eps > 0.3 * self.prev:
                if capped and self.inner_cap < 16 * self.base:
                    self.inner_cap *= 2
                    self.hist = []
                elif self.aspect > 100:
                    self.cap = max(100, self.aspect // 2)
            reach = subspace_reach(ritz, self.Emin, self.Emax, 0.8 if not (eps < 1e-2) else 0.95) if M > 0 else None
            self.aspect = (policy_pick(self.Emin, self.Emax, self.ne, self.q, self.rt, self.cap, reach, ritz[:M], 2 * self.aspect)
                           if reach is not None else min(self.aspect, self.cap))
        self.next_rtol = self.rt
        if self.tol > 0 and math.isfinite(eps) and eps > self.tol:
            self.next_rtol = min(0.3, max(self.rt, 0.32 * self.tol / eps))
        self.prev = eps
