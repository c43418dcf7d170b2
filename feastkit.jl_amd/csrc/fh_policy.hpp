// fh_policy.hpp -- host-side policy of the INEXACT FEAST mode (pure C++17, no HIP): what a host shim needs besides the
// kernels to make the default call `feast(A, B, interval; M0, fpm)` fast on large sparse input, kept under the C ABI so that
// the Julia shim of INTEGRATION.md calls it instead of re-porting it (feasthip_policy_* in include/feasthip.h).
//
// Not in the reference.  With exact solves the contraction of FEAST's subspace iteration per refinement loop is the filter
// ratio rho(lambda_{M0+1}) / rho(lambda_inside) -- 1e-5 and better for a 16-point Gauss rule on the circle.  With inner
// solves that only reduce the residual by `inner_rtol` per loop the contraction is max(filter ratio, ~inner_rtol): a filter
// sharper than inner_rtol is paid for (its nodes sit next to the real axis, where the shifted systems are worst
// conditioned) and never used.  The policy picks the ellipse ratio fpm[18] -- the reference's own parameter,
// src/core/feast_parameters.jl:232-247, src/core/feast_tools.jl:212-284 -- that minimises the predicted work
//        a^-0.6 / ln(1 / max(filter_ratio(a), inner_rtol))
// (a^-0.6: measured fall of the Krylov iterations per loop with the ratio), re-evaluated every loop at the reach of the
// current subspace, with two safeguards (a loop that contracts by less than 0.3: double the inner iteration cap when nodes
// stopped at it, else halve the ratio; an outer residual that has not halved over two loops: double the cap), the
// set-aside rule for noise pairs, and the inner tolerance of the LAST loop (no more reduction than the outer tolerance
// still needs).  tests/policy_reference.py restates the rules in numpy; tests/test_host_logic.py pins this file against it.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace fh_policy {

static const int kAspectCandidates[] = {100, 150, 200, 300, 400, 600, 800, 1200, 1600, 2400, 3200, 4000, 5000, 6000, 8000};
static const int kNumCandidates = (int)(sizeof(kAspectCandidates) / sizeof(kAspectCandidates[0]));

// Gauss-Legendre nodes (ascending) and weights on (-1, 1): Newton on P_n with the Tricomi start; unique, so this equals
// FastGaussQuadrature.gausslegendre / numpy leggauss to rounding
static inline void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w) {
    x.assign(n, 0.0); w.assign(n, 0.0);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(pi * (i + 0.75) / (n + 0.5));
        double pp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            const double dz = p1 / pp;
            z -= dz;
            if (std::fabs(dz) < 1e-16) break;
        }
        // one more evaluation of the derivative at the converged node
        double p1 = 1.0, p2 = 0.0;
        for (int j = 1; j <= n; ++j) { const double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j; }
        pp = n * (z * p1 - p2) / (z * z - 1.0);
        x[i] = -z; x[n - 1 - i] = z;
        w[i] = w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

// half contour of the unit interval (-1, 1): Gauss (quadrature 0) or trapezoid (1) nodes on the ellipse of ratio aspect / 100
// (src/core/feast_tools.jl:212-284 with Emid = 0, r = 1)
static inline void unit_contour(int ne, int quadrature, int aspect100, std::vector<std::complex<double>>& Z, std::vector<std::complex<double>>& W) {
    const double pi = 3.14159265358979323846, aspect = aspect100 * 0.01;
    std::vector<double> x, w;
    if (quadrature == 0) gauss_legendre(ne, x, w);
    Z.resize(ne); W.resize(ne);
    for (int e = 0; e < ne; ++e) {
        double theta, fac;
        if (quadrature == 0) { theta = -pi / 2 * x[e] + pi / 2; fac = 0.25 * w[e]; }
        else { theta = pi - (pi / ne) / 2 - (pi / ne) * e; fac = 1.0 / (2 * ne); }
        Z[e] = std::complex<double>(std::cos(theta), aspect * std::sin(theta));
        W[e] = fac * std::complex<double>(aspect * std::cos(theta), std::sin(theta));
    }
}

// rho(lambda) = Re sum_e 2 w_e / (z_e - lambda): the rational filter of the half contour with the real projection
static inline double filter_value(const std::vector<std::complex<double>>& Z, const std::vector<std::complex<double>>& W, double lam) {
    double s = 0.0;
    for (size_t e = 0; e < Z.size(); ++e) s += (2.0 * W[e] / (Z[e] - lam)).real();
    return s;
}

struct filter_table {
    std::vector<std::complex<double>> Z, W;
    std::vector<double> d, env;            // outer envelope E(d) = max_{d' >= d} |rho(d')| on a log grid 1 <= d <= 60
};
static const int kGrid = 4000;

static inline const filter_table& table(int ne, int quadrature, int aspect100) {
    static std::map<std::tuple<int, int, int>, filter_table> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(ne, quadrature, aspect100);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    filter_table t;
    unit_contour(ne, quadrature, aspect100, t.Z, t.W);
    t.d.resize(kGrid); t.env.resize(kGrid);
    const double l60 = std::log(60.0);
    for (int k = 0; k < kGrid; ++k) t.d[k] = std::exp(l60 * k / (kGrid - 1));
    double run = 0.0;
    for (int k = kGrid - 1; k >= 0; --k) { run = std::max(run, std::fabs(filter_value(t.Z, t.W, t.d[k]))); t.env[k] = run; }
    return cache.emplace(key, std::move(t)).first->second;
}

// upper envelope of |rho| over |lambda - Emid| >= d_rel * r, divided by the smallest |rho| over the wanted eigenvalues
// (`inside`: their current Ritz values; none: 65 points of the interval, whose ends carry rho = 1/2)
static inline double filter_ratio(double Emin, double Emax, int ne, int quadrature, int aspect100, double d_rel, const double* inside, int n_inside) {
    const filter_table& t = table(ne, quadrature, aspect100);
    const double r = 0.5 * (Emax - Emin), mid = Emin + r;
    const double dq = std::max(d_rel, 1.0);
    int idx = (int)(std::lower_bound(t.d.begin(), t.d.end(), dq) - t.d.begin());
    if (idx > kGrid - 1) idx = kGrid - 1;
    const double out = t.env[idx];
    double inn = 1e300;
    if (!inside || n_inside <= 0) {
        for (int k = 0; k < 65; ++k) inn = std::min(inn, std::fabs(filter_value(t.Z, t.W, -1.0 + 2.0 * k / 64.0)));
    } else {
        for (int k = 0; k < n_inside; ++k) inn = std::min(inn, std::fabs(filter_value(t.Z, t.W, (inside[k] - mid) / r)));
    }
    return out / std::max(inn, 1e-300);
}

// how far the current subspace reaches beyond the interval, in half widths from the midpoint: the `quantile` point of the
// distances of the guard Ritz values (those outside [Emin, Emax]); < 0 when there are no guards
static inline double subspace_reach(const double* ritz, int n, double Emin, double Emax, double quantile) {
    const double r = 0.5 * (Emax - Emin), mid = Emin + r;
    std::vector<double> g;
    for (int i = 0; i < n; ++i) if (ritz[i] < Emin || ritz[i] > Emax) g.push_back(std::fabs(ritz[i] - mid));
    if (g.empty()) return -1.0;
    std::sort(g.begin(), g.end());
    const int k = std::min((int)g.size() - 1, (int)(quantile * g.size()));
    return g[k] / r;
}

// fpm[18] minimising the predicted work at subspace reach d_rel, among the candidates <= cap (and <= limit when limit > 0)
static inline int pick(double Emin, double Emax, int ne, int quadrature, double inner_rtol, int cap, double d_rel, const double* inside,
                       int n_inside, int limit) {
    int best = 100;
    double best_cost = -1.0;
    for (int k = 0; k < kNumCandidates; ++k) {
        const int a = kAspectCandidates[k];
        if (a > cap || (limit > 0 && a > limit)) continue;
        const double c = std::max(filter_ratio(Emin, Emax, ne, quadrature, a, d_rel, inside, n_inside), inner_rtol);
        if (c >= 0.5) continue;
        const double cost = std::pow((double)a, -0.6) / std::log(1.0 / c);
        if (best_cost < 0.0 || cost < best_cost) { best = a; best_cost = cost; }
    }
    return best;
}

}   // namespace fh_policy
