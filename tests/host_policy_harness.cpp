// CPU harness for the host policy of the inexact FEAST mode (feastkit.jl_amd/csrc/fh_policy.hpp), built by
// tests/test_ingest_sanitizer.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all  (GPU sanitizers are not
// available on this pool; this half of the library is pure C++).  Checks, over random and degenerate inputs:
//   * Gauss-Legendre nodes / weights: symmetric, ascending, weights sum to 2, exact for polynomials up to degree 2n - 1;
//   * the filter of the half contour with the real projection is ~1 inside, 1/2 at the ends, decays outside;
//   * filter_ratio is monotone in the reach, invariant under shift / scaling, finite for every candidate ratio;
//   * subspace_reach ignores Ritz values inside, returns < 0 without guards, handles n = 0;
//   * pick() always returns a candidate within cap / limit, 100 when nothing qualifies.
// Usage: host_policy_harness [cases] [seed]  -> prints "ok <cases>" or aborts.
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../feastkit.jl_amd/csrc/fh_policy.hpp"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); std::abort(); } } while (0)

int main(int argc, char** argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 200;
    std::mt19937_64 rng(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 20260515ull);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int n = 1; n <= 40; ++n) {
        std::vector<double> x, w;
        fh_policy::gauss_legendre(n, x, w);
        double sw = 0.0;
        for (int i = 0; i < n; ++i) {
            sw += w[i];
            CHECK(w[i] > 0.0 && std::fabs(x[i] + x[n - 1 - i]) < 1e-14 && std::fabs(w[i] - w[n - 1 - i]) < 1e-14);
            if (i) CHECK(x[i] > x[i - 1]);
        }
        CHECK(std::fabs(sw - 2.0) < 1e-13);
        for (int d = 0; d <= 2 * n - 1; d += 1) {            // integral of t^d over (-1, 1)
            double q = 0.0;
            for (int i = 0; i < n; ++i) q += w[i] * std::pow(x[i], d);
            const double exact = (d % 2) ? 0.0 : 2.0 / (d + 1);
            CHECK(std::fabs(q - exact) < 1e-12);
        }
    }
    for (int ne : {4, 8, 16, 24})
        for (int quad : {0, 1}) {
            std::vector<std::complex<double>> Z, W;
            fh_policy::unit_contour(ne, quad, 100, Z, W);
            const double tol = (quad == 0 && ne >= 8) ? 1e-5 : 0.2;
            CHECK(std::fabs(fh_policy::filter_value(Z, W, 0.0) - 1.0) < tol);
            if (quad == 0 && ne >= 8) {
                CHECK(std::fabs(fh_policy::filter_value(Z, W, 1.0) - 0.5) < 1e-5 && std::fabs(fh_policy::filter_value(Z, W, -1.0) - 0.5) < 1e-5);
                CHECK(std::fabs(fh_policy::filter_value(Z, W, 3.0)) < 1e-3);
            }
        }
    for (int c = 0; c < cases; ++c) {
        const double Emin = -3.0 + 6.0 * u(rng), Emax = Emin + 1e-3 + 4.0 * u(rng);
        const int ne = 2 + (int)(u(rng) * 30), quad = u(rng) < 0.5 ? 0 : 1;
        const int a = fh_policy::kAspectCandidates[(int)(u(rng) * fh_policy::kNumCandidates) % fh_policy::kNumCandidates];
        double prev = 1e300;
        for (double d : {1.0, 1.3, 2.0, 5.0, 30.0, 100.0}) {
            const double r1 = fh_policy::filter_ratio(Emin, Emax, ne, quad, a, d, nullptr, 0);
            CHECK(std::isfinite(r1) && r1 >= 0.0 && r1 <= prev * (1 + 1e-12));
            prev = r1;
            const double r2 = fh_policy::filter_ratio(10.0 * Emin + 7.0, 10.0 * Emax + 7.0, ne, quad, a, d, nullptr, 0);
            CHECK(std::fabs(r1 - r2) <= 1e-6 * r1 + 1e-300);
        }
        const int n = (int)(u(rng) * 70);
        std::vector<double> ritz(n);
        int guards = 0;
        const double r = 0.5 * (Emax - Emin), mid = Emin + r;
        for (int i = 0; i < n; ++i) {
            ritz[i] = mid + r * (u(rng) < 0.5 ? (2.0 * u(rng) - 1.0) * 0.999 : (u(rng) < 0.5 ? -1.0 : 1.0) * (1.001 + 5.0 * u(rng)));
            if (ritz[i] < Emin || ritz[i] > Emax) ++guards;
        }
        const double reach = fh_policy::subspace_reach(ritz.data(), n, Emin, Emax, u(rng));
        if (guards == 0) CHECK(reach < 0.0); else CHECK(reach > 1.0 && reach < 6.1);
        CHECK(fh_policy::subspace_reach(nullptr, 0, Emin, Emax, 0.8) < 0.0);
        const int cap = fh_policy::kAspectCandidates[(int)(u(rng) * fh_policy::kNumCandidates) % fh_policy::kNumCandidates];
        const int limit = u(rng) < 0.5 ? 0 : 100 + (int)(u(rng) * 9000);
        const double rt = std::pow(10.0, -3.0 * u(rng) - 0.5);
        const int got = fh_policy::pick(Emin, Emax, ne, quad, rt, cap, 1.0 + 3.0 * u(rng), n ? ritz.data() : nullptr, 0, limit);
        bool member = false;
        for (int k = 0; k < fh_policy::kNumCandidates; ++k) member |= fh_policy::kAspectCandidates[k] == got;
        CHECK(member && (got == 100 || (got <= cap && (limit <= 0 || got <= limit))));
        CHECK(fh_policy::pick(Emin, Emax, ne, quad, 0.9, cap, 1.5, nullptr, 0, 0) == 100);        // nothing contracts below 0.5
    }
    std::printf("ok %d\n", cases);
    return 0;
}
