// host_mf_harness.cpp -- CPU harness of the multifrontal plan (feastkit.jl_amd/csrc/fh_mf.hpp): builds plans for grid,
// random and degenerate patterns, checks the plan's invariants, and EXECUTES the plan on the CPU exactly as the device code
// does (padded fronts, partial pivoting inside the fully-summed block, extend-add through the plan's maps, forward /
// backward substitution through the tree) against the residual of the solve.  Test infrastructure: built and run by
// tests/test_ingest_sanitizer.py under -fsanitize=address,undefined.
//   host_mf_harness            the fixed cases
//   host_mf_harness stats nx ny nz leaf     plan statistics of a 3-D 7-point grid (no numeric run)
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include "../feastkit.jl_amd/csrc/fh_mf.hpp"

typedef std::complex<double> cd;

struct csr { int N; std::vector<int> rowptr, col; std::vector<double> a, b; bool bident; };

static csr grid3d(int nx, int ny, int nz, bool bident) {
    csr M; M.N = nx * ny * nz; M.bident = bident;
    M.rowptr.assign(M.N + 1, 0);
    auto id = [&](int i, int j, int k) { return (i * ny + j) * nz + k; };
    for (int i = 0; i < nx; ++i) for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) {
        const int r = id(i, j, k);
        auto put = [&](int c, double v) { M.col.push_back(c); M.a.push_back(v); M.b.push_back(c == r ? 1.0 + 0.1 * v : 0.1 * v); };
        if (i > 0) put(id(i - 1, j, k), -1.0);
        if (j > 0) put(id(i, j - 1, k), -1.0);
        if (k > 0) put(id(i, j, k - 1), -1.0);
        put(r, 6.0);
        if (k + 1 < nz) put(id(i, j, k + 1), -1.0);
        if (j + 1 < ny) put(id(i, j + 1, k), -1.0);
        if (i + 1 < nx) put(id(i + 1, j, k), -1.0);
        M.rowptr[r + 1] = (int)M.col.size();
    }
    return M;
}

// random pattern (unsymmetric, some rows without a stored diagonal), diagonally weighted so that z B - A is well conditioned
static csr random_pattern(int N, int per_row, unsigned seed, bool bident, bool drop_diag) {
    std::mt19937 rng(seed);
    csr M; M.N = N; M.bident = bident;
    M.rowptr.assign(N + 1, 0);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int i = 0; i < N; ++i) {
        std::vector<int> cols;
        for (int q = 0; q < per_row; ++q) {
            int span = (q % 3 == 0) ? N : 40;                   // mostly local couplings, a few long ones
            int c = (i + (int)(rng() % (unsigned)(2 * span + 1)) - span) % N;
            if (c < 0) c += N;
            cols.push_back(c);
        }
        if (!(drop_diag && i % 7 == 3)) cols.push_back(i);
        std::sort(cols.begin(), cols.end());
        cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
        for (int c : cols) { M.col.push_back(c); M.a.push_back(c == i ? 8.0 + U(rng) : U(rng)); M.b.push_back(c == i ? 2.0 + 0.1 * U(rng) : 0.1 * U(rng)); }
        M.rowptr[i + 1] = (int)M.col.size();
    }
    return M;
}

static int fail(const char* what, int code) { std::printf("FAIL %s (%d)\n", what, code); return 1; }

static int check_plan(const csr& M, const fh_mf::plan& P) {
    const int N = M.N;
    std::vector<int> seen(N, 0);
    for (int i = 0; i < N; ++i) { if (P.perm[i] < 0 || P.perm[i] >= N || seen[P.perm[i]]++) return fail("perm", i); if (P.iperm[P.perm[i]] != i) return fail("iperm", i); }
    size_t piv = 0;
    for (size_t f = 0; f < P.fronts.size(); ++f) {
        const fh_mf::front& F = P.fronts[f];
        if (F.piv0 != (int)piv) return fail("piv0", (int)f);
        piv += F.npiv;
        if (F.npiv <= 0) return fail("npiv", (int)f);
        const fh_mf::group& G = P.groups[F.group];
        if (G.fronts[F.slot] != (int)f) return fail("slot", (int)f);
        if (F.npiv > G.np || F.nbnd > G.nb || G.np % 32 || G.n != G.np + G.nb) return fail("geometry", (int)f);
        for (int q = 0; q < F.nbnd; ++q) {
            const int u = P.bnd[F.bnd_off + q];
            if (u < F.piv0 + F.npiv || u >= N) return fail("bnd range", (int)f);
            if (q && P.bnd[F.bnd_off + q - 1] >= u) return fail("bnd order", (int)f);
            if (F.parent >= 0) {
                const fh_mf::front& Pf = P.fronts[F.parent];
                const fh_mf::group& Gp = P.groups[Pf.group];
                const int r = P.rel[F.bnd_off + q];
                if (r < 0 || r >= Gp.n) return fail("rel range", (int)f);
                const int idx = r < Gp.np ? Pf.piv0 + r : P.bnd[Pf.bnd_off + r - Gp.np];
                if (r < Gp.np ? r >= Pf.npiv : r - Gp.np >= Pf.nbnd) return fail("rel pad", (int)f);
                if (idx != u) return fail("rel target", (int)f);
                if (Pf.group <= F.group) return fail("group order", (int)f);
            }
        }
        if (F.parent < 0 && F.nbnd) return fail("root boundary", (int)f);
        for (int c : F.child) if (c >= 0 && (P.fronts[c].parent != (int)f || c >= (int)f)) return fail("child", (int)f);
    }
    if ((int)piv != N) return fail("pivot count", (int)piv);
    {   // work arena: regions whose lifetimes overlap must not (a group lives from the start of its level -- the groups of a
        // level may run side by side -- to the end of the level of its last parent group)
        const int ng = (int)P.groups.size();
        std::vector<int> first(ng), last(ng);
        for (int g = 0; g < ng; ++g) {
            int a = g, b = g;
            while (a > 0 && P.groups[a - 1].height == P.groups[g].height) --a;
            while (b + 1 < ng && P.groups[b + 1].height == P.groups[g].height) ++b;
            first[g] = a; last[g] = b;
        }
        std::vector<int> until(ng);
        for (int g = 0; g < ng; ++g) {
            int u = g;
            for (int f : P.groups[g].fronts) if (P.fronts[f].parent >= 0) u = std::max(u, P.fronts[P.fronts[f].parent].group);
            until[g] = last[u];
        }
        for (int a = 0; a < ng; ++a)
            for (int b = a + 1; b < ng; ++b) {
                if (first[b] > until[a]) continue;                      // b's level starts after a was given back
                const size_t a0 = P.groups[a].work_off, a1 = a0 + P.groups[a].work_per * P.groups[a].fronts.size();
                const size_t b0 = P.groups[b].work_off, b1 = b0 + P.groups[b].work_per * P.groups[b].fronts.size();
                if (a0 < b1 && b0 < a1) return fail("arena overlap", a * 1000 + b);
                if (a1 > P.work_elems || b1 > P.work_elems) return fail("arena size", a);
            }
    }
    // every CSR entry assembled exactly once; identity diagonals exactly once
    std::vector<int> hits(M.col.size(), 0);
    int diag = 0;
    for (size_t g = 0; g < P.groups.size(); ++g) {
        const fh_mf::group& G = P.groups[g];
        for (size_t q = G.asm_begin; q < G.asm_end; ++q) {
            const int d = P.asm_dst[q] < 0 ? ~P.asm_dst[q] : P.asm_dst[q];
            if (P.asm_dst[q] < 0) ++diag;
            if ((size_t)d >= G.work_per * G.fronts.size()) return fail("asm range", (int)q);
            const size_t within = (size_t)d % G.work_per;
            if (within >= (size_t)G.n * G.n) return fail("asm in inverse area", (int)q);
            if (P.asm_src[q] >= 0) hits[P.asm_src[q]]++;
            else if (!M.bident) return fail("asm src", (int)q);
        }
    }
    for (size_t k = 0; k < hits.size(); ++k) if (hits[k] != 1) return fail("asm coverage", (int)k);
    if (M.bident && diag != N) return fail("identity diagonal", diag);
    return 0;
}

// the plan executed on the CPU for one shift z; returns the relative residual of (zB - A) x = rhs
static double execute(const csr& M, const fh_mf::plan& P, cd z, int* info) {
    const int N = M.N;
    std::vector<std::vector<cd>> W(P.groups.size());            // per group: fronts x (n x n), column-major
    std::vector<std::vector<int>> PV(P.groups.size());
    *info = 0;
    for (size_t g = 0; g < P.groups.size(); ++g) {
        const fh_mf::group& G = P.groups[g];
        const int n = G.n, np = G.np;
        W[g].assign(G.work_per * G.fronts.size(), cd(0, 0));
        PV[g].assign((size_t)np * G.fronts.size(), 0);
        for (size_t s = 0; s < G.fronts.size(); ++s)                // identity on the pad pivots
            for (int p = P.fronts[G.fronts[s]].npiv; p < np; ++p) W[g][s * G.work_per + p + (size_t)p * n] = cd(1, 0);
        for (size_t q = G.asm_begin; q < G.asm_end; ++q) {
            const bool dg = P.asm_dst[q] < 0;
            const int d = dg ? ~P.asm_dst[q] : P.asm_dst[q];
            const int k = P.asm_src[q];
            cd v(0, 0);
            if (k >= 0) v = (M.bident ? cd(0, 0) : z * M.b[k]) - M.a[k];
            if (dg) v += z;
            W[g][d] += v;
        }
        for (int side = 0; side < 2; ++side)
            for (int c : G.kids[side]) {
                const fh_mf::front& C = P.fronts[c];
                const fh_mf::group& Gc = P.groups[C.group];
                const cd* S = W[C.group].data() + (size_t)C.slot * Gc.work_per;
                cd* F = W[g].data() + (size_t)P.fronts[C.parent].slot * G.work_per;
                for (int j = 0; j < C.nbnd; ++j)
                    for (int i = 0; i < C.nbnd; ++i)
                        F[P.rel[C.bnd_off + i] + (size_t)P.rel[C.bnd_off + j] * n] += S[(Gc.np + i) + (size_t)(Gc.np + j) * Gc.n];
            }
        for (size_t s = 0; s < G.fronts.size(); ++s) {              // partial LU, pivots among rows < np
            cd* F = W[g].data() + s * G.work_per;
            int* pv = PV[g].data() + s * np;
            for (int k = 0; k < np; ++k) {
                int p = k; double best = -1.0;
                for (int i = k; i < np; ++i) { const double m = std::fabs(F[i + (size_t)k * n].real()) + std::fabs(F[i + (size_t)k * n].imag()); if (m > best) { best = m; p = i; } }
                if (!(best > 0.0)) { *info = 1; return 1e300; }
                pv[k] = p;
                if (p != k) for (int c = 0; c < n; ++c) std::swap(F[k + (size_t)c * n], F[p + (size_t)c * n]);
                const cd inv = cd(1, 0) / F[k + (size_t)k * n];
                for (int i = k + 1; i < n; ++i) F[i + (size_t)k * n] *= inv;
                for (int c = k + 1; c < n; ++c) {
                    const cd u = F[k + (size_t)c * n];
                    if (u == cd(0, 0)) continue;
                    for (int i = k + 1; i < n; ++i) F[i + (size_t)c * n] -= F[i + (size_t)k * n] * u;
                }
            }
        }
    }
    // solve one right-hand side
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<cd> rhs(N), x(N);
    for (int i = 0; i < N; ++i) rhs[i] = cd(U(rng), U(rng));
    std::vector<std::vector<cd>> Y(P.groups.size());
    for (size_t g = 0; g < P.groups.size(); ++g) {                  // forward
        const fh_mf::group& G = P.groups[g];
        const int n = G.n, np = G.np;
        Y[g].assign((size_t)n * G.fronts.size(), cd(0, 0));
        for (size_t s = 0; s < G.fronts.size(); ++s) {
            const fh_mf::front& F = P.fronts[G.fronts[s]];
            for (int p = 0; p < F.npiv; ++p) Y[g][s * n + p] = rhs[P.perm[F.piv0 + p]];
        }
        for (int side = 0; side < 2; ++side)
            for (int c : G.kids[side]) {
                const fh_mf::front& C = P.fronts[c];
                const fh_mf::group& Gc = P.groups[C.group];
                for (int i = 0; i < C.nbnd; ++i)
                    Y[g][(size_t)P.fronts[C.parent].slot * n + P.rel[C.bnd_off + i]] += Y[C.group][(size_t)C.slot * Gc.n + Gc.np + i];
            }
        for (size_t s = 0; s < G.fronts.size(); ++s) {
            const cd* F = W[g].data() + s * G.work_per;
            cd* y = Y[g].data() + s * n;
            const int* pv = PV[g].data() + s * np;
            for (int k = 0; k < np; ++k) if (pv[k] != k) std::swap(y[k], y[pv[k]]);
            for (int k = 0; k < np; ++k) for (int i = k + 1; i < n; ++i) y[i] -= F[i + (size_t)k * n] * y[k];
        }
    }
    for (size_t g = P.groups.size(); g-- > 0;) {                    // backward
        const fh_mf::group& G = P.groups[g];
        const int n = G.n, np = G.np;
        for (size_t s = 0; s < G.fronts.size(); ++s) {
            const fh_mf::front& F = P.fronts[G.fronts[s]];
            const cd* A = W[g].data() + s * G.work_per;
            cd* y = Y[g].data() + s * n;
            if (F.parent >= 0) {
                const fh_mf::front& Pf = P.fronts[F.parent];
                const fh_mf::group& Gp = P.groups[Pf.group];
                for (int i = 0; i < F.nbnd; ++i) y[np + i] = Y[Pf.group][(size_t)Pf.slot * Gp.n + P.rel[F.bnd_off + i]];
            }
            for (int i = F.nbnd; i < G.nb; ++i) y[np + i] = cd(0, 0);
            for (int k = np - 1; k >= 0; --k) {
                cd sum = y[k];
                for (int c = k + 1; c < n; ++c) sum -= A[k + (size_t)c * n] * y[c];
                y[k] = sum / A[k + (size_t)k * n];
            }
            for (int p = 0; p < F.npiv; ++p) x[P.perm[F.piv0 + p]] = y[p];
        }
    }
    double rn = 0.0, bn = 0.0;
    for (int i = 0; i < N; ++i) {
        cd r = rhs[i];
        for (int k = M.rowptr[i]; k < M.rowptr[i + 1]; ++k) {
            const cd s = (M.bident ? (M.col[k] == i ? z : cd(0, 0)) : z * M.b[k]) - M.a[k];
            r -= s * x[M.col[k]];
        }
        if (M.bident) {
            bool has = false;
            for (int k = M.rowptr[i]; k < M.rowptr[i + 1]; ++k) if (M.col[k] == i) has = true;
            if (!has) r -= z * x[i];
        }
        rn += std::norm(r); bn += std::norm(rhs[i]);
    }
    return std::sqrt(rn / bn);
}

static void print_stats(const char* name, const csr& M, const fh_mf::plan& P) {
    int leaves = 0;
    for (const fh_mf::front& F : P.fronts) if (F.child[0] < 0) ++leaves;
    std::printf("%s: N %d  fronts %zu (%d leaves)  groups %zu  max front %d (np %d)  flops %.3e (exact %.3e)  store %.3f GB  work %.3f GB  rhs rows %zu\n",
                name, M.N, P.fronts.size(), leaves, P.groups.size(), P.max_n, P.max_np, P.flops, P.flops_exact, P.store_elems * 16.0 / 1e9,
                P.work_elems * 16.0 / 1e9, P.rhs_rows);
}

int main(int argc, char** argv) {
    if (argc >= 6 && !std::strcmp(argv[1], "stats")) {
        const csr M = grid3d(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), false);
        fh_mf::plan P;
        const int rc = fh_mf::make_plan(M.N, M.rowptr, M.col, false, atoi(argv[5]), P);
        if (rc) return fail("make_plan", rc);
        if (check_plan(M, P)) return 1;
        print_stats("grid", M, P);
        if (argc >= 7)
            for (size_t g = 0; g < P.groups.size(); ++g)
                std::printf("  group %2zu: height %2d  fronts %4zu  np %4d  nb %4d  n %4d  flops %.2e\n", g, P.groups[g].height, P.groups[g].fronts.size(),
                            P.groups[g].np, P.groups[g].nb, P.groups[g].n, P.groups[g].flops);
        return 0;
    }
    struct tc { const char* name; csr M; int leaf; };
    std::vector<tc> cases;
    cases.push_back({"grid 12x10x8", grid3d(12, 10, 8, false), 24});
    cases.push_back({"grid 9x9x9 B=I", grid3d(9, 9, 9, true), 16});
    cases.push_back({"grid 30x1x1 (path)", grid3d(30, 1, 1, false), 8});
    cases.push_back({"grid 40x25x1", grid3d(40, 25, 1, false), 32});
    cases.push_back({"random 700", random_pattern(700, 4, 1, false, false), 32});
    cases.push_back({"random 900 B=I, missing diagonals", random_pattern(900, 3, 2, true, true), 24});
    cases.push_back({"random dense-ish 200", random_pattern(200, 40, 3, false, false), 16});
    cases.push_back({"tiny 5", random_pattern(5, 2, 4, false, false), 8});
    {   // two disconnected grids
        csr a = grid3d(6, 5, 4, false), b = grid3d(4, 4, 4, false);
        csr M; M.N = a.N + b.N; M.bident = false; M.rowptr = a.rowptr; M.col = a.col; M.a = a.a; M.b = a.b;
        for (int i = 0; i < b.N; ++i) {
            for (int k = b.rowptr[i]; k < b.rowptr[i + 1]; ++k) { M.col.push_back(b.col[k] + a.N); M.a.push_back(b.a[k]); M.b.push_back(b.b[k]); }
            M.rowptr.push_back((int)M.col.size());
        }
        cases.push_back({"two components", M, 16});
    }
    int bad = 0;
    for (tc& c : cases) {
        fh_mf::plan P;
        const int rc = fh_mf::make_plan(c.M.N, c.M.rowptr, c.M.col, c.M.bident, c.leaf, P);
        if (rc) { bad += fail(c.name, rc); continue; }
        if (check_plan(c.M, P)) { std::printf("  in %s\n", c.name); ++bad; continue; }
        int info = 0;
        const double res = execute(c.M, P, cd(0.7, 0.9), &info);
        print_stats(c.name, c.M, P);
        std::printf("   relative residual %.2e  info %d\n", res, info);
        if (info || !(res < 1e-11)) { std::printf("FAIL residual in %s\n", c.name); ++bad; }
    }
    {   // malformed input: column out of range
        csr M = grid3d(3, 3, 3, false);
        M.col[5] = 1000;
        fh_mf::plan P;
        if (fh_mf::make_plan(M.N, M.rowptr, M.col, false, 8, P) == 0) { std::printf("FAIL accepted an out-of-range column\n"); ++bad; }
    }
    std::printf(bad ? "FAILED (%d)\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
