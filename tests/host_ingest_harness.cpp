// CPU fuzz harness for the host side of feasthip_set_csr (feastkit.jl_amd/csrc/fh_ingest.hpp), built by
// tests/test_ingest_sanitizer.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all.
// Random pencils in every input form the C ABI accepts (CSR / CSC, 0- / 1-based, unsorted rows, duplicate entries, empty
// rows, B on a different pattern or absent, real and complex values) go through fh_prepare_csr with and without the
// row-block renumbering; every output array is checked against a dense accumulation of the input:
//   * the union pattern carries exactly A and B (duplicates summed), under the renumbering perm (a bijection);
//   * kl / ku are the band widths in caller order; every row's first entry has its largest column;
//   * the chunk-of-8 rows hold the same entries, padded with (own row, 0, 0);
//   * row blocks tile [0, N) in pieces of at most R rows; the LDS slot of every nonzero names its column;
//   * the reverse Cuthill-McKee order of the band plan is a bijection and fh_bandwidth counts what it should;
//   * malformed pointers / indices are rejected, never read out of bounds (the sanitizers watch).
// Usage: host_ingest_harness [cases] [seed]   -> prints "ok <cases>" or aborts.
#include <cassert>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

struct c2 { double x, y; };
static inline c2 fh_ing_zero(c2) { return c2{0.0, 0.0}; }
static inline c2 fh_ing_add(c2 a, c2 b) { return c2{a.x + b.x, a.y + b.y}; }
#include "../feastkit.jl_amd/csrc/fh_ingest.hpp"

static int g_case = 0;
#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed line %d: %s (case %d)\n", __LINE__, #cond, g_case); std::abort(); } } while (0)

static double re(double v) { return v; }
static double im(double) { return 0.0; }
static double re(c2 v) { return v.x; }
static double im(c2 v) { return v.y; }
static void mk(double& v, double a, double) { v = a; }
static void mk(c2& v, double a, double b) { v = c2{a, b}; }

template <typename VT>
struct input {
    std::vector<int64_t> ptr, idx;
    std::vector<VT> val;
    std::map<std::pair<int, int>, std::complex<double>> dense;      // summed entries (row, col) in caller terms
};

template <typename VT>
static input<VT> random_matrix(std::mt19937_64& rng, int N, int base, int storage, double density) {
    input<VT> in;
    std::uniform_real_distribution<double> u(0.0, 1.0);
    std::vector<std::vector<std::pair<int, VT>>> lines(N);             // per row (CSR) or per column (CSC)
    for (int l = 0; l < N; ++l) {
        if (u(rng) < 0.1) continue;                                      // an empty line now and then
        const int cnt = 1 + (int)(u(rng) * density * N);
        for (int k = 0; k < cnt; ++k) {
            const int o = (int)(u(rng) * N) % N;
            VT v; mk(v, u(rng) - 0.5, u(rng) - 0.5);
            lines[l].push_back({o, v});
            if (u(rng) < 0.15) lines[l].push_back({o, v});              // duplicate entry (summed by the ingest)
        }
    }
    in.ptr.assign(N + 1, base);
    for (int l = 0; l < N; ++l) {
        for (auto& e : lines[l]) {                                       // deliberately NOT sorted
            in.idx.push_back(e.first + base);
            in.val.push_back(e.second);
            const int r = storage == 0 ? l : e.first, c = storage == 0 ? e.first : l;
            in.dense[{r, c}] += std::complex<double>(re(e.second), im(e.second));
        }
        in.ptr[l + 1] = (int64_t)in.idx.size() + base;
    }
    return in;
}

template <typename VT>
static void one_case(std::mt19937_64& rng, bool real_chunks) {
    std::uniform_real_distribution<double> u(0.0, 1.0);
    const int R = 16, EXT = 12;                                          // small blocks so that small matrices renumber
    const int N = 1 + (int)(u(rng) * 90);
    const int base = u(rng) < 0.5 ? 0 : 1, storage = u(rng) < 0.5 ? 0 : 1;
    const bool hasB = u(rng) < 0.7;
    const int reorder = (int)(u(rng) * 3);
    input<VT> A = random_matrix<VT>(rng, N, base, storage, 0.08), B;
    if (hasB) B = random_matrix<VT>(rng, N, base, storage, 0.05);
    fh_prepared<VT> P;
    std::string err;
    const int rc = fh_prepare_csr<VT>(N, base, storage, (int64_t)A.idx.size(), A.ptr.data(), A.idx.data(), A.val.data(),
                                      hasB ? (int64_t)B.idx.size() : 0, hasB ? B.ptr.data() : nullptr, hasB ? B.idx.data() : nullptr,
                                      hasB ? B.val.data() : nullptr, reorder, R, EXT, real_chunks, P, err);
    CHECK(rc == 0);
    CHECK((int)P.rowptr.size() == N + 1 && P.rowptr[0] == 0 && P.rowptr[N] == (int)P.col.size());
    CHECK(P.av.size() == P.col.size() && (hasB ? P.bv.size() == P.col.size() : P.bv.empty()));
    // renumbering: a bijection, blocks tile [0, N) with at most R rows each
    std::vector<int> old_of(N);
    for (int i = 0; i < N; ++i) old_of[i] = i;
    if (!P.perm.empty()) {
        CHECK((int)P.perm.size() == N);
        std::vector<int> seen(N, 0);
        for (int i = 0; i < N; ++i) { CHECK(P.perm[i] >= 0 && P.perm[i] < N); seen[P.perm[i]]++; old_of[i] = P.perm[i]; }
        for (int i = 0; i < N; ++i) CHECK(seen[i] == 1);
        CHECK(P.blk_start.front() == 0 && P.blk_start.back() == N);
        for (size_t b = 0; b + 1 < P.blk_start.size(); ++b) CHECK(P.blk_start[b + 1] > P.blk_start[b] && P.blk_start[b + 1] - P.blk_start[b] <= R);
        CHECK(P.lcol.size() == P.col.size() && P.ext_ptr.size() == P.blk_start.size());
    } else {
        CHECK(reorder == 0 || N < 2 * R || (reorder == 1 && !(N >= 4 * R && P.kl + P.ku > 512)));
    }
    // union pattern == dense accumulation of the input, every (row, col) once; largest column first
    std::map<std::pair<int, int>, std::pair<std::complex<double>, std::complex<double>>> got;
    int kl = 0, ku = 0;
    for (int i = 0; i < N; ++i) {
        CHECK(P.rowptr[i + 1] >= P.rowptr[i]);
        for (int k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k) {
            CHECK(P.col[k] >= 0 && P.col[k] < N);
            CHECK(P.col[k] <= P.col[P.rowptr[i]]);
            const std::pair<int, int> key{old_of[i], old_of[P.col[k]]};
            CHECK(!got.count(key));
            got[key] = {std::complex<double>(re(P.av[k]), im(P.av[k])), hasB ? std::complex<double>(re(P.bv[k]), im(P.bv[k])) : std::complex<double>(0, 0)};
            kl = std::max(kl, key.first - key.second);
            ku = std::max(ku, key.second - key.first);
        }
    }
    CHECK(kl == P.kl && ku == P.ku);
    for (auto& e : A.dense) { CHECK(got.count(e.first)); CHECK(std::abs(got[e.first].first - e.second) <= 1e-14); }
    if (hasB) for (auto& e : B.dense) { CHECK(got.count(e.first)); CHECK(std::abs(got[e.first].second - e.second) <= 1e-14); }
    for (auto& e : got) {
        const auto ia = A.dense.find(e.first);
        CHECK(std::abs(e.second.first - (ia == A.dense.end() ? std::complex<double>(0, 0) : ia->second)) <= 1e-14);
        if (hasB) {
            const auto ib = B.dense.find(e.first);
            CHECK(std::abs(e.second.second - (ib == B.dense.end() ? std::complex<double>(0, 0) : ib->second)) <= 1e-14);
            CHECK(ia != A.dense.end() || ib != B.dense.end());
        } else {
            CHECK(ia != A.dense.end());
        }
    }
    // chunk-of-8 rows
    if (real_chunks) {
        CHECK((int)P.rp8.size() == N + 1 && P.rp8[0] == 0);
        CHECK(P.col8.size() >= (size_t)P.rp8[N] * 8 && P.a8.size() == P.col8.size() && (hasB ? P.b8.size() == P.col8.size() : P.b8.empty()));
        for (int i = 0; i < N; ++i) {
            const int len = P.rowptr[i + 1] - P.rowptr[i];
            CHECK(P.rp8[i + 1] - P.rp8[i] == (len + 7) / 8);
            for (int q = 0; q < (P.rp8[i + 1] - P.rp8[i]) * 8; ++q) {
                const size_t w = (size_t)P.rp8[i] * 8 + q;
                if (q < len) {
                    CHECK(P.col8[w] == P.col[P.rowptr[i] + q] && P.a8[w] == re(P.av[P.rowptr[i] + q]));
                    if (hasB) CHECK(P.b8[w] == re(P.bv[P.rowptr[i] + q]));
                } else {
                    CHECK(P.col8[w] == i && P.a8[w] == 0.0);
                    if (hasB) CHECK(P.b8[w] == 0.0);
                }
            }
        }
    } else {
        CHECK(P.rp8.empty() && P.col8.empty());
    }
    // LDS slots name the right columns
    if (!P.perm.empty()) {
        for (size_t b = 0; b + 1 < P.blk_start.size(); ++b) {
            const int r0 = P.blk_start[b], r1 = P.blk_start[b + 1];
            CHECK(P.ext_ptr[b + 1] - P.ext_ptr[b] <= EXT);
            for (int k = P.rowptr[r0]; k < P.rowptr[r1]; ++k) {
                const int c = P.col[k];
                const unsigned short s = P.lcol[k];
                if (c >= r0 && c < r1) CHECK(s == c - r0);
                else if (s != 0xFFFF) { CHECK(s >= R && s - R < P.ext_ptr[b + 1] - P.ext_ptr[b]); CHECK(P.ext_idx[P.ext_ptr[b] + (s - R)] == c); }
            }
        }
    }
    // band plan of the direct solver: reverse Cuthill-McKee gives a bijection, fh_bandwidth agrees with a direct count
    {
        std::vector<int> rperm;
        fh_rcm(N, P.rowptr, P.col, rperm);
        CHECK((int)rperm.size() == N);
        std::vector<int> ip(N, -1);
        for (int i = 0; i < N; ++i) { CHECK(rperm[i] >= 0 && rperm[i] < N && ip[rperm[i]] == -1); ip[rperm[i]] = i; }
        int bkl = 0, bku = 0, ckl = 0, cku = 0;
        fh_bandwidth(N, P.rowptr, P.col, ip.data(), bkl, bku);
        for (int i = 0; i < N; ++i)
            for (int k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k) { ckl = std::max(ckl, ip[i] - ip[P.col[k]]); cku = std::max(cku, ip[P.col[k]] - ip[i]); }
        CHECK(bkl == ckl && bku == cku && bkl < N && bku < N);
    }
    // malformed input is rejected: an index beyond N, a pointer beyond nnz, a pointer below the base
    if (!A.idx.empty()) {
        input<VT> bad = A;
        bad.idx[(size_t)(u(rng) * bad.idx.size()) % bad.idx.size()] = N + base + (int)(u(rng) * 5);
        fh_prepared<VT> Q;
        CHECK(fh_prepare_csr<VT>(N, base, storage, (int64_t)bad.idx.size(), bad.ptr.data(), bad.idx.data(), bad.val.data(), 0, nullptr,
                                 nullptr, nullptr, 0, R, EXT, real_chunks, Q, err) == 1);
        bad = A;
        bad.ptr[N] = (int64_t)bad.idx.size() + base + 3;
        CHECK(fh_prepare_csr<VT>(N, base, storage, (int64_t)bad.idx.size(), bad.ptr.data(), bad.idx.data(), bad.val.data(), 0, nullptr,
                                 nullptr, nullptr, 0, R, EXT, real_chunks, Q, err) == 1);
        bad = A;
        bad.ptr[0] = base - 1;
        CHECK(fh_prepare_csr<VT>(N, base, storage, (int64_t)bad.idx.size(), bad.ptr.data(), bad.idx.data(), bad.val.data(), 0, nullptr,
                                 nullptr, nullptr, 0, R, EXT, real_chunks, Q, err) == 1);
        // non-monotone pointers with every value in range: an interior pointer pushed past its successor (overlapping
        // ranges visit more than nnz entries), in both storages
        if (N >= 3) {
            bad = A;
            const int i = 1 + (int)(u(rng) * (N - 2));                   // 1 .. N-2
            bad.ptr[i] = bad.ptr[i + 1] + 1 + (int64_t)(u(rng) * 3);
            if (bad.ptr[i] - base <= (int64_t)bad.idx.size())
                CHECK(fh_prepare_csr<VT>(N, base, storage, (int64_t)bad.idx.size(), bad.ptr.data(), bad.idx.data(), bad.val.data(), 0,
                                         nullptr, nullptr, nullptr, 0, R, EXT, real_chunks, Q, err) == 1);
        }
        // the same defect in B only
        if (hasB && N >= 3 && !B.idx.empty()) {
            input<VT> badB = B;
            badB.ptr[1] = (int64_t)badB.idx.size() + base;
            badB.ptr[2] = base;
            CHECK(fh_prepare_csr<VT>(N, base, storage, (int64_t)A.idx.size(), A.ptr.data(), A.idx.data(), A.val.data(),
                                     (int64_t)badB.idx.size(), badB.ptr.data(), badB.idx.data(), badB.val.data(), 0, R, EXT,
                                     real_chunks, Q, err) == 2);
        }
    }
}

// The advisor's case (round 3): N = 3, nnz = 5, ptr = {0, 5, 0, 5}, every index 2 -- all values in range, the column
// ranges overlap, and the CSC counting transpose wrote fill[r]++ past out.idx / out.val (ASan: heap-buffer-overflow).
static void overlapping_ranges_case() {
    for (int storage = 0; storage < 2; ++storage)
        for (int base = 0; base < 2; ++base) {
            std::vector<int64_t> ptr = {0 + base, 5 + base, 0 + base, 5 + base}, idx(5, 2 + base);
            std::vector<double> val(5, 1.0);
            fh_prepared<double> Q;
            std::string err;
            CHECK(fh_prepare_csr<double>(3, base, storage, 5, ptr.data(), idx.data(), val.data(), 0, nullptr, nullptr, nullptr, 0, 16,
                                         12, true, Q, err) == 1);
            // as B beside a well-formed A
            std::vector<int64_t> pa = {0 + base, 1 + base, 2 + base, 3 + base}, ia = {0 + base, 1 + base, 2 + base};
            std::vector<double> va(3, 2.0);
            CHECK(fh_prepare_csr<double>(3, base, storage, 3, pa.data(), ia.data(), va.data(), 5, ptr.data(), idx.data(), val.data(), 0,
                                         16, 12, true, Q, err) == 2);
        }
}

int main(int argc, char** argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 300;
    std::mt19937_64 rng(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 20260515ull);
    overlapping_ranges_case();
    for (g_case = 0; g_case < cases; ++g_case) {
        if (g_case & 1) one_case<double>(rng, true);
        else one_case<c2>(rng, false);
    }
    std::printf("ok %d\n", cases);
    return 0;
}
