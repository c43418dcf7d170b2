// fh_ingest.hpp -- host-side ingest of a sparse pencil (pure C++17, no HIP): everything feasthip_set_csr does before the
// upload.  Kept free of device headers so that tests/host_ingest_harness.cpp can compile it with gcc under
// AddressSanitizer / UBSan and fuzz it on the CPU (tests/test_ingest_sanitizer.py); fh_api.hip includes the same file.
//
//   (CSR | CSC, 0- | 1-based, unsorted rows, duplicates)  ->  0-based CSR rows sorted by column          to_csr0
//   A, B -> union pattern, duplicates summed, band widths kl / ku                                         fh_prepare_csr
//   optional renumbering into row blocks (LDS-window SpMM) + LDS slot of every nonzero                    fh_block_partition
//   largest column first in every row (far gather of k_spmm)
//   chunk-of-8 padded rows (k_spmm_row; real values only)
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#ifndef FH_INGEST_STORAGE_CSR
#define FH_INGEST_STORAGE_CSR 0      // == FEASTHIP_STORAGE_CSR (include/feasthip.h)
#endif

template <typename VT>
struct host_csr {
    std::vector<int64_t> ptr, idx;
    std::vector<VT> val;
};

// value arithmetic the ingest needs: zero and sum (double here; fh_api.hip adds the complex overloads before including)
static inline double fh_ing_zero(double) { return 0.0; }
static inline double fh_ing_add(double a, double b) { return a + b; }

// Convert (CSR|CSC, base) input to 0-based CSR with sorted rows.
template <typename VT>
static bool to_csr0(int64_t N, int index_base, int storage, int64_t nnz, const int64_t* ptr, const int64_t* idx,
                    const VT* val, host_csr<VT>& out) {
    // pointers: start at the base, never decrease, end at nnz -- overlapping ranges (ptr = {0, 5, 0, 5}) would visit more
    // than nnz entries and run the counting transpose past its output arrays
    if (N < 0 || nnz < 0 || ptr[0] != index_base || ptr[N] - index_base != nnz) return false;
    for (int64_t i = 0; i < N; ++i)
        if (ptr[i + 1] < ptr[i]) return false;
    for (int64_t k = 0; k < nnz; ++k)
        if (idx[k] - index_base < 0 || idx[k] - index_base >= N) return false;
    out.ptr.assign(N + 1, 0);
    out.idx.resize(nnz);
    out.val.resize(nnz);
    if (storage == FH_INGEST_STORAGE_CSR) {
        for (int64_t i = 0; i <= N; ++i) out.ptr[i] = ptr[i] - index_base;
        for (int64_t k = 0; k < nnz; ++k) { out.idx[k] = idx[k] - index_base; out.val[k] = val[k]; }
    } else {
        // CSC -> CSR: counting transpose (entry (r, c) stored in column c)
        for (int64_t k = 0; k < nnz; ++k) out.ptr[idx[k] - index_base + 1]++;
        for (int64_t i = 0; i < N; ++i) out.ptr[i + 1] += out.ptr[i];
        std::vector<int64_t> fill(out.ptr.begin(), out.ptr.end() - 1);
        for (int64_t c = 0; c < N; ++c)
            for (int64_t k = ptr[c] - index_base; k < ptr[c + 1] - index_base; ++k) {
                int64_t r = idx[k] - index_base;
                int64_t o = fill[r]++;
                out.idx[o] = c;
                out.val[o] = val[k];
            }
    }
    // sort each row by column (insertion sort: rows are short / mostly sorted)
    for (int64_t i = 0; i < N; ++i) {
        int64_t a = out.ptr[i], b = out.ptr[i + 1];
        bool sorted = true;
        for (int64_t k = a + 1; k < b; ++k) if (out.idx[k] < out.idx[k - 1]) { sorted = false; break; }
        if (sorted) continue;
        std::vector<std::pair<int64_t, VT>> row;
        row.reserve(b - a);
        for (int64_t k = a; k < b; ++k) row.push_back({out.idx[k], out.val[k]});
        std::stable_sort(row.begin(), row.end(), [](const std::pair<int64_t, VT>& x, const std::pair<int64_t, VT>& y) { return x.first < y.first; });
        for (int64_t k = a; k < b; ++k) { out.idx[k] = row[k - a].first; out.val[k] = row[k - a].second; }
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// Row blocks for the LDS-window SpMM (fh_sparse.hip).  That kernel stages the X rows a block of at most FH_SPMM_R
// consecutive matrix rows touches into LDS; it only pays when most of a block's column indices fall inside the block and
// the rest hit few distinct rows.  Ingest therefore renumbers the unknowns by recursive bisection of the union pattern:
// breadth-first levels from a pseudo-peripheral vertex of the part, cut at a multiple of FH_SPMM_R near the middle, both
// halves again, until a part fits one block.  On the 7-point pattern of cfg 3 a 128-row block reaches 133 distinct
// outside rows on average (greedy graph growing, tried first: 281).  The permutation never leaves the library: panels are
// permuted when they cross the C ABI (column-major in caller order <-> row-major panel in block order), every reduction is
// order independent, results are for the matrix as the caller defined it.  perm[new] = old; blk_start: first row of every
// block (+ N).
// ---------------------------------------------------------------------------------------
static void fh_block_partition(int64_t N, const std::vector<int>& rowptr_in, const std::vector<int>& col_in, int R,
                               std::vector<int>& perm, std::vector<int>& blk_start) {
    // The traversal needs an UNDIRECTED graph: on a structurally unsymmetric pattern (general pencils) a breadth-first
    // search along the stored direction only reaches part of a component, the level order of a part then listed some
    // vertices twice and the permutation was not a bijection (found by tests/host_ingest_harness.cpp under ASan: heap
    // overflow while re-packing the rows).  Adjacency = pattern united with its transpose.
    std::vector<int> rowptr(N + 1, 0), col;
    {
        std::vector<int> deg(N, 0);
        for (int64_t i = 0; i < N; ++i)
            for (int k = rowptr_in[i]; k < rowptr_in[i + 1]; ++k) { deg[i]++; if (col_in[k] != i) deg[col_in[k]]++; }
        for (int64_t i = 0; i < N; ++i) rowptr[i + 1] = rowptr[i] + deg[i];
        col.resize(rowptr[N]);
        std::vector<int> fill(rowptr.begin(), rowptr.end() - 1);
        for (int64_t i = 0; i < N; ++i)
            for (int k = rowptr_in[i]; k < rowptr_in[i + 1]; ++k) {
                const int j = col_in[k];
                col[fill[i]++] = j;
                if (j != i) col[fill[j]++] = (int)i;
            }
    }
    perm.resize(N);
    for (int64_t i = 0; i < N; ++i) perm[i] = (int)i;
    blk_start.clear();
    std::vector<int> part(N, 0), seen(N, -1), buf, probe;
    buf.reserve(N);
    int next_pid = 1, next_id = 0;
    // breadth-first order of part `pid` from `start` (marks: seen[v] = id), appended to out; returns the last vertex reached
    auto bfs = [&](int start, int pid, int id, std::vector<int>& out) -> int {
        const size_t first = out.size();
        seen[start] = id;
        out.push_back(start);
        for (size_t head = first; head < out.size(); ++head) {
            const int v = out[head];
            for (int k = rowptr[v]; k < rowptr[v + 1]; ++k) {
                const int u = col[k];
                if (part[u] == pid && seen[u] != id) { seen[u] = id; out.push_back(u); }
            }
        }
        return out.back();
    };
    struct range { int64_t lo, hi; int pid; };
    std::vector<range> stack, leaves;
    stack.push_back({0, N, 0});
    while (!stack.empty()) {
        const range r = stack.back();
        stack.pop_back();
        const int64_t n = r.hi - r.lo;
        if (n <= R) { leaves.push_back(r); continue; }
        // level order of the part: every connected component from a pseudo-peripheral vertex (the far end of a probe BFS)
        const int id_probe = next_id++, id_order = next_id++;
        buf.clear();
        for (int64_t i = r.lo; i < r.hi; ++i) {
            const int v = perm[i];
            if (seen[v] == id_order) continue;                 // placed with an earlier component
            probe.clear();
            const int far = bfs(v, r.pid, id_probe, probe);
            bfs(far, r.pid, id_order, buf);
        }
        for (int64_t i = 0; i < n; ++i) perm[r.lo + i] = buf[i];
        int64_t left = ((n / 2 + R / 2) / R) * R;              // cut at a multiple of R near the middle: full blocks
        left = std::max<int64_t>(R, std::min<int64_t>(left, n - 1));
        const int p1 = next_pid++, p2 = next_pid++;
        for (int64_t i = 0; i < left; ++i) part[perm[r.lo + i]] = p1;
        for (int64_t i = left; i < n; ++i) part[perm[r.lo + i]] = p2;
        stack.push_back({r.lo + left, r.hi, p2});
        stack.push_back({r.lo, r.lo + left, p1});
    }
    std::sort(leaves.begin(), leaves.end(), [](const range& a, const range& b) { return a.lo < b.lo; });
    for (const range& r : leaves) blk_start.push_back((int)r.lo);
    blk_start.push_back((int)N);
}


// ---------------------------------------------------------------------------------------
// Band plan of the sparse DIRECT solver (fh_banded.hip / fh_dense.hip: blocked band LU on the dense LU kernels).
// fh_bandwidth: lower / upper band widths of a pattern under a renumbering (iperm[old] = new; null: as stored).
// fh_rcm: reverse Cuthill-McKee order of the pattern united with its transpose -- every connected component from a
// pseudo-peripheral vertex (the far end of a breadth-first probe), neighbours taken by increasing degree, the whole order
// reversed.  perm[new] = old.
// ---------------------------------------------------------------------------------------
static void fh_bandwidth(int64_t N, const std::vector<int>& rowptr, const std::vector<int>& col, const int* iperm, int& kl, int& ku) {
    kl = 0; ku = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int bi = iperm ? iperm[i] : (int)i;
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int bj = iperm ? iperm[col[k]] : col[k];
            kl = std::max(kl, bi - bj);
            ku = std::max(ku, bj - bi);
        }
    }
}

static void fh_rcm(int64_t N, const std::vector<int>& rowptr_in, const std::vector<int>& col_in, std::vector<int>& perm) {
    std::vector<int> rowptr(N + 1, 0), col;
    {
        std::vector<int> deg(N, 0);
        for (int64_t i = 0; i < N; ++i)
            for (int k = rowptr_in[i]; k < rowptr_in[i + 1]; ++k) if (col_in[k] != i) { deg[i]++; deg[col_in[k]]++; }
        for (int64_t i = 0; i < N; ++i) rowptr[i + 1] = rowptr[i] + deg[i];
        col.resize(rowptr[N]);
        std::vector<int> fill(rowptr.begin(), rowptr.end() - 1);
        for (int64_t i = 0; i < N; ++i)
            for (int k = rowptr_in[i]; k < rowptr_in[i + 1]; ++k) {
                const int j = col_in[k];
                if (j != i) { col[fill[i]++] = j; col[fill[j]++] = (int)i; }
            }
    }
    // (a symmetric pattern lists every edge twice here; duplicates only cost time in the traversal)
    std::vector<int> deg(N);
    for (int64_t i = 0; i < N; ++i) deg[i] = rowptr[i + 1] - rowptr[i];
    std::vector<int> mark(N, -1), order, probe, nb;
    order.reserve(N);
    int id = 0;
    auto bfs = [&](int start, int tag, std::vector<int>& out, bool by_degree) -> int {
        const size_t first = out.size();
        mark[start] = tag;
        out.push_back(start);
        for (size_t head = first; head < out.size(); ++head) {
            const int v = out[head];
            nb.clear();
            for (int k = rowptr[v]; k < rowptr[v + 1]; ++k) {
                const int u = col[k];
                if (mark[u] != tag && mark[u] != -2) { mark[u] = tag; nb.push_back(u); }
            }
            if (by_degree) std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] < deg[b]; });
            for (int u : nb) out.push_back(u);
        }
        return out.back();
    };
    for (int64_t s = 0; s < N; ++s) {
        if (mark[s] == -2) continue;                       // -2: placed
        probe.clear();
        const int far = bfs((int)s, id++, probe, false);
        // the probe only marks with a fresh tag; the ordering pass marks placed vertices for good
        const size_t first = order.size();
        bfs(far, id++, order, true);
        for (size_t q = first; q < order.size(); ++q) mark[order[q]] = -2;
    }
    perm.assign(order.rbegin(), order.rend());
}

template <typename VT>
struct fh_prepared {
    std::vector<int> rowptr, col;              // union pattern, 0-based, (optionally renumbered,) largest column first per row
    std::vector<VT> av, bv;                    // A and B on the union pattern (bv empty: B = I)
    int kl = 0, ku = 0;                        // band widths of the union pattern in CALLER order
    std::vector<int> perm, blk_start;          // renumbering perm[new] = old and row blocks (empty: none)
    std::vector<int> ext_ptr, ext_idx;         // per row block: the distinct outside rows it touches (<= ext_max kept)
    std::vector<unsigned short> lcol;          // LDS slot of every nonzero (renumbered matrices only)
    std::vector<int> rp8, col8;                // chunk-of-8 rows (real values only): chunk range of row i = [rp8[i], rp8[i+1])
    std::vector<double> a8, b8;
};

// reorder: 0 none, 1 renumber wide patterns (N >= 4 R and kl + ku > 512), 2 renumber whenever N >= 2 R.
// Returns 0, or 1 (malformed A), 2 (malformed B), 3 (nnz of the union pattern exceeds int32); `err` says which.
template <typename VT>
static int fh_prepare_csr(int64_t N, int index_base, int storage, int64_t nnzA, const int64_t* ptrA, const int64_t* idxA,
                          const VT* valA, int64_t nnzB, const int64_t* ptrB, const int64_t* idxB, const VT* valB,
                          int reorder, int R, int ext_max, bool real_chunks, fh_prepared<VT>& out, std::string& err) {
    host_csr<VT> A, B;
    if (!to_csr0<VT>(N, index_base, storage, nnzA, ptrA, idxA, valA, A)) { err = "malformed A (pointer/index out of range)"; return 1; }
    const bool hasB = ptrB != nullptr;
    if (hasB && !to_csr0<VT>(N, index_base, storage, nnzB, ptrB, idxB, valB, B)) { err = "malformed B (pointer/index out of range)"; return 2; }
    std::vector<int>& rowptr = out.rowptr;
    std::vector<int>& col = out.col;
    std::vector<VT>& av = out.av;
    std::vector<VT>& bv = out.bv;
    rowptr.assign(N + 1, 0); col.clear(); av.clear(); bv.clear();
    col.reserve(nnzA + (hasB ? nnzB : 0));
    av.reserve(col.capacity());
    if (hasB) bv.reserve(col.capacity());
    // union pattern, duplicates summed
    for (int64_t i = 0; i < N; ++i) {
        int64_t ka = A.ptr[i], ea = A.ptr[i + 1];
        int64_t kb = hasB ? B.ptr[i] : 0, eb = hasB ? B.ptr[i + 1] : 0;
        while (ka < ea || kb < eb) {
            int64_t ca = ka < ea ? A.idx[ka] : INT64_MAX;
            int64_t cb = kb < eb ? B.idx[kb] : INT64_MAX;
            int64_t c = std::min(ca, cb);
            VT a = fh_ing_zero(VT()), b = fh_ing_zero(VT());
            while (ka < ea && A.idx[ka] == c) { a = fh_ing_add(a, A.val[ka]); ++ka; }
            while (kb < eb && B.idx[kb] == c) { b = fh_ing_add(b, B.val[kb]); ++kb; }
            col.push_back((int)c);
            av.push_back(a);
            if (hasB) bv.push_back(b);
        }
        if (col.size() > (size_t)INT32_MAX) { err = "nnz exceeds int32"; return 3; }
        rowptr[i + 1] = (int)col.size();
    }
    out.kl = out.ku = 0;
    for (int64_t i = 0; i < N; ++i)
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            out.kl = std::max(out.kl, (int)(i - col[k]));
            out.ku = std::max(out.ku, (int)(col[k] - i));
        }
    // Renumber into row blocks when the pattern is too wide for the banded LU anyway (that solver needs caller order)
    out.perm.clear(); out.blk_start.clear(); out.ext_ptr.clear(); out.ext_idx.clear(); out.lcol.clear();
    if ((reorder == 1 && N >= 4 * R && out.kl + out.ku > 512) || (reorder == 2 && N >= 2 * R)) {
        fh_block_partition(N, rowptr, col, R, out.perm, out.blk_start);
        const std::vector<int>& perm = out.perm;
        std::vector<int> inv(N);
        for (int64_t i = 0; i < N; ++i) inv[perm[i]] = (int)i;
        std::vector<int> rp2(N + 1, 0), col2(col.size());
        std::vector<VT> av2(av.size()), bv2(bv.size());
        std::vector<std::pair<int, int>> row;
        for (int64_t i = 0; i < N; ++i) {
            const int o = perm[i];
            row.clear();
            for (int k = rowptr[o]; k < rowptr[o + 1]; ++k) row.push_back({inv[col[k]], k});
            std::sort(row.begin(), row.end());
            int w = rp2[i];
            for (auto& e : row) {
                col2[w] = e.first; av2[w] = av[e.second];
                if (hasB) bv2[w] = bv[e.second];
                ++w;
            }
            rp2[i + 1] = w;
        }
        rowptr.swap(rp2); col.swap(col2); av.swap(av2); bv.swap(bv2);
    }
    // The nonzero with the LARGEST column index goes first in its row.  k_spmm sweeps the rows in ascending order, so that
    // is the X row nobody has touched yet (the one gather of a row that comes from HBM, not from L2): the kernel
    // issues it one row step ahead (fh_sparse.hip).  The order inside a row means nothing else to any kernel.
    for (int64_t i = 0; i < N; ++i) {
        int kmax = rowptr[i];
        for (int k = rowptr[i] + 1; k < rowptr[i + 1]; ++k) if (col[k] > col[kmax]) kmax = k;
        if (kmax != rowptr[i]) {
            std::swap(col[kmax], col[rowptr[i]]); std::swap(av[kmax], av[rowptr[i]]);
            if (hasB) std::swap(bv[kmax], bv[rowptr[i]]);
        }
    }
    // chunk-of-8 copy of the rows for k_spmm_row (fh_sparse.hip): one s_load_dwordx8 / x16 per chunk, no tail tests
    out.rp8.clear(); out.col8.clear(); out.a8.clear(); out.b8.clear();
    if (real_chunks) {
        out.rp8.assign(N + 1, 0);
        for (int64_t i = 0; i < N; ++i) out.rp8[i + 1] = out.rp8[i] + (rowptr[i + 1] - rowptr[i] + 7) / 8;
        const size_t n8 = (size_t)out.rp8[N] * 8;
        out.col8.assign(std::max<size_t>(8, n8), 0);
        out.a8.assign(std::max<size_t>(8, n8), 0.0);
        if (hasB) out.b8.assign(std::max<size_t>(8, n8), 0.0);
        const double* avd = (const double*)av.data();
        const double* bvd = (const double*)bv.data();
        for (int64_t i = 0; i < N; ++i) {
            size_t w = (size_t)out.rp8[i] * 8;
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k, ++w) { out.col8[w] = col[k]; out.a8[w] = avd[k]; if (hasB) out.b8[w] = bvd[k]; }
            for (; w < (size_t)out.rp8[i + 1] * 8; ++w) out.col8[w] = (int)i;
        }
    }
    if (!out.perm.empty()) {
        // LDS slots of every nonzero
        const std::vector<int>& blk_start = out.blk_start;
        const int nb = (int)blk_start.size() - 1;
        out.ext_ptr.assign(nb + 1, 0);
        out.lcol.assign(col.size(), 0);
        std::vector<int> uniq;
        for (int b = 0; b < nb; ++b) {
            const int r0 = blk_start[b], r1 = blk_start[b + 1];
            uniq.clear();
            for (int k = rowptr[r0]; k < rowptr[r1]; ++k) if (col[k] < r0 || col[k] >= r1) uniq.push_back(col[k]);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            if ((int)uniq.size() > ext_max) uniq.resize(ext_max);       // the rest is gathered from global memory
            for (int k = rowptr[r0]; k < rowptr[r1]; ++k) {
                const int c = col[k];
                if (c >= r0 && c < r1) out.lcol[k] = (unsigned short)(c - r0);
                else {
                    auto it = std::lower_bound(uniq.begin(), uniq.end(), c);
                    out.lcol[k] = (it != uniq.end() && *it == c) ? (unsigned short)(R + (it - uniq.begin())) : (unsigned short)0xFFFF;
                }
            }
            out.ext_idx.insert(out.ext_idx.end(), uniq.begin(), uniq.end());
            out.ext_ptr[b + 1] = (int)out.ext_idx.size();
        }
    }
    return 0;
}
