// fh_mf.hpp -- symbolic phase of the multifrontal sparse direct solver (pure C++17, no HIP; fuzzed on the CPU by
// tests/host_mf_harness.cpp under ASan/UBSan).
//
// Role in the reference: `lu(z*B - A)` of a SparseMatrixCSC is UMFPACK, a fill-reducing multifrontal factorisation
// (src/sparse/feast_sparse.jl:334-342, src/core/feast_backend_utils.jl:174-179).  The band LU of fh_dense.hip fills the
// whole band after reverse Cuthill-McKee (cfg 3: 2.77 GB and 7.4e11 flop per quadrature node); this plan confines the fill to
// the fronts of a nested-dissection elimination tree.
//
//   ordering   recursive vertex bisection of pattern(A) U pattern(A)^T U pattern(B): breadth-first level structure from a
//              pseudo-peripheral vertex, the separator is the level that balances the halves with the fewest vertices,
//              thinned by moving separator vertices without a neighbour on one side to the other side (George & Liu's
//              automatic nested dissection).  Subsets of <= leaf vertices are leaves.  Elimination order = post order of
//              the separator tree (left subtree, right subtree, separator).
//   fronts     one dense front per tree vertex: npiv fully-summed unknowns (the separator / leaf) + nbnd boundary unknowns
//              (the part of the ancestors' separators the subtree is connected to, by fill).  struct(f) = (adj(V_f) U
//              struct(children)) \ V_f, computed bottom-up on sorted index lists.
//   groups     fronts of equal tree height and similar size share one PADDED geometry (np pivots, nb boundary rows, order
//              n = np + nb): pad pivots are identity rows/columns, pad boundary rows are zero.  One group = one batch of
//              equal-size dense matrices for the batched LU / substitution kernels (fronts x quadrature nodes).
//   maps       assembly list (CSR entry -> position in its front: the front that owns the earlier-eliminated index of the
//              entry), extend-add maps (boundary row of a child -> row of its parent's padded front), pivot -> unknown.
//
// Pivoting is partial pivoting INSIDE the fully-summed block of a front (rows < np); a pivot is never taken from a boundary
// row (no delayed pivots).  For z off the real axis and a definite B the leading blocks of z B - A are nonsingular (the
// field of values of z B - A misses zero), which is the FEAST case; the caller checks the factorisation's info flags and the
// residual of the solve and falls back to the band LU otherwise.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace fh_mf {

struct front {
    int piv0 = 0, npiv = 0, nbnd = 0;          // pivots = new indices [piv0, piv0 + npiv)
    int parent = -1, child[2] = {-1, -1};
    int height = 0, group = -1, slot = 0;      // slot: position inside the group
    size_t bnd_off = 0;                        // struct(f) = bnd[bnd_off .. bnd_off + nbnd), sorted new indices; rel[bnd_off + i] = row of the PARENT's padded front boundary row i lands in
};

struct group {
    int height = 0, np = 0, nb = 0, n = 0;     // padded geometry: np (multiple of 32) pivots, nb boundary rows, n = np + nb
    int inv128 = 0;                            // 128-block inverses kept (large pivot blocks only)
    std::vector<int> fronts;
    size_t work_off = 0, work_per = 0;         // work arena (elements per quadrature node): fronts x [n x n matrix + 32-block inverses]
    size_t store_off = 0, store_per = 0;       // factor store: fronts x [L block column n x np | inverses | U12 np x nb]
    size_t piv_off = 0;                        // pivot store (ints): fronts x [np pivots | np row permutation]
    size_t rhs_off = 0;                        // substitution panels (rows): fronts x n
    size_t asm_begin = 0, asm_end = 0;         // assembly list range
    std::vector<int> kids[2];                  // fronts whose parent is in this group: first children, second children (two
                                               // children of one parent add into the same entries: one after the other)
    double flops = 0.0;                        // 8 x complex multiply-adds of the padded partial factorisation, all fronts
};

struct plan {
    int N = 0;
    std::vector<int> perm, iperm;              // perm[new] = old, iperm[old] = new
    std::vector<front> fronts;                 // post order: children before parents
    std::vector<int> bnd;
    std::vector<group> groups;                 // by height (leaves first)
    std::vector<int> asm_dst;                  // element offset inside the group's work region of ONE front-set (slot * work_per + row + col * n); ~x: diagonal of an identity B
    std::vector<int> asm_src;                  // CSR entry (position in the caller's col / value arrays), -1: none (B = I diagonal without an A entry)
    std::vector<int> rel;                      // see front::bnd_off
    size_t work_elems = 0, store_elems = 0, piv_ints = 0, rhs_rows = 0;
    double flops = 0.0;                        // padded, per quadrature node
    double flops_exact = 0.0;                  // unpadded
    int max_n = 0, max_np = 0;
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline double plan_store_slack() { const char* e = getenv("FH_MF_STORE_SLACK"); const double v = e ? atof(e) : 0.0; return v >= 1.0 ? v : 1.25; }
static inline size_t inv32_elems(int np) { return (size_t)(np / 32) * 2 * 32 * 32; }
static inline size_t inv128_elems(int np) { return (size_t)((np + 127) / 128) * 2 * 128 * 128; }

// real flops of a partial LU of order n eliminating p columns (complex arithmetic: 8 per multiply-add)
static inline double partial_lu_flops(double n, double p) {
    // sum_{k=0}^{p-1} (n-k-1)^2 multiply-adds ~ p n^2 - p^2 n + p^3/3
    return 8.0 * (p * n * n - p * p * n + p * p * p / 3.0);
}

struct builder {
    int N;
    std::vector<int> xadj, adj;                // symmetrised pattern without the diagonal
    std::vector<int> mark, mark2, level, queue;
    int stamp = 0;
    int leaf;
    plan* P;
    int next_piv = 0;

    int new_front(const std::vector<int>& verts, int c0, int c1) {
        front f;
        f.piv0 = next_piv; f.npiv = (int)verts.size();
        f.child[0] = c0; f.child[1] = c1;
        for (int v : verts) { P->perm[next_piv] = v; P->iperm[v] = next_piv; ++next_piv; }
        const int id = (int)P->fronts.size();
        f.height = 0;
        for (int c : {c0, c1}) if (c >= 0) { P->fronts[c].parent = id; f.height = std::max(f.height, P->fronts[c].height + 1); }
        P->fronts.push_back(f);
        return id;
    }

    // breadth-first level structure of the subgraph induced by `verts` (mark == tag), all components, from `start`;
    // fills queue (visit order) and level[], returns the number of levels
    int bfs_levels(const std::vector<int>& verts, int tag, int start) {
        const int seen = ++stamp;
        queue.clear();
        int nlev = 0;
        size_t next_seed = 0;
        int seed = start;
        while (true) {
            size_t head = queue.size();
            level[seed] = nlev; mark2[seed] = seen; queue.push_back(seed);
            for (; head < queue.size(); ++head) {
                const int v = queue[head];
                for (int k = xadj[v]; k < xadj[v + 1]; ++k) {
                    const int u = adj[k];
                    if (mark[u] == tag && mark2[u] != seen) { mark2[u] = seen; level[u] = level[v] + 1; queue.push_back(u); }
                }
            }
            nlev = level[queue.back()] + 1;
            if (queue.size() == verts.size()) break;
            while (mark2[verts[next_seed]] == seen) ++next_seed;     // another component: its levels follow
            seed = verts[next_seed];
        }
        return nlev;
    }
    int dissect(std::vector<int>& verts) {
        const int nv = (int)verts.size();
        if (nv <= leaf) return new_front(verts, -1, -1);
        const int tag = ++stamp;
        for (int v : verts) mark[v] = tag;
        // pseudo-peripheral start: far end of a probe, twice
        int start = verts[0];
        for (int pass = 0; pass < 2; ++pass) {
            bfs_levels(verts, tag, start);
            // the last level's vertex of least degree (inside the subset the degrees are close; the global one will do)
            const int last = level[queue.back()];
            int best = queue.back();
            for (size_t q = queue.size(); q-- > 0 && level[queue[q]] == last;)
                if (xadj[queue[q] + 1] - xadj[queue[q]] < xadj[best + 1] - xadj[best]) best = queue[q];
            start = best;
        }
        const int nlev = bfs_levels(verts, tag, start);
        if (nlev < 3) return new_front(verts, -1, -1);            // no level separates anything: one dense front
        std::vector<int> cnt(nlev + 1, 0);
        for (int v : verts) cnt[level[v] + 1]++;
        for (int l = 0; l < nlev; ++l) cnt[l + 1] += cnt[l];     // cnt[l] = vertices in levels < l
        int bestk = -1;
        double bestscore = 0.0;
        for (int relax = 0; relax < 2 && bestk < 0; ++relax) {
            for (int k = 1; k + 1 < nlev; ++k) {
                const int left = cnt[k], right = nv - cnt[k + 1], sz = cnt[k + 1] - cnt[k];
                if (left == 0 || right == 0) continue;
                if (!relax && std::min(left, right) < 0.25 * nv) continue;
                const double score = sz * (1.0 + 2.0 * std::abs(left - right) / (double)nv);
                if (bestk < 0 || score < bestscore) { bestk = k; bestscore = score; }
            }
        }
        if (bestk < 0) return new_front(verts, -1, -1);
        // side: 0 left, 1 separator, 2 right (kept in level[] as -1 / -2 / -3 to reuse the array)
        std::vector<int> L, S, R;
        for (int v : verts) {
            if (level[v] < bestk) L.push_back(v);
            else if (level[v] > bestk) R.push_back(v);
            else S.push_back(v);
        }
        // thinning: a separator vertex without a neighbour in R joins L; then one without a neighbour in L joins R
        {
            std::vector<int> S2;
            for (int v : S) {
                bool touches_right = false;
                for (int k = xadj[v]; k < xadj[v + 1]; ++k) { const int u = adj[k]; if (mark[u] == tag && level[u] > bestk) { touches_right = true; break; } }
                if (touches_right) S2.push_back(v); else { L.push_back(v); level[v] = bestk - 1; }
            }
            S.swap(S2);
            S2.clear();
            for (int v : S) {
                bool touches_left = false;
                for (int k = xadj[v]; k < xadj[v + 1]; ++k) { const int u = adj[k]; if (mark[u] == tag && level[u] < bestk) { touches_left = true; break; } }
                if (touches_left) S2.push_back(v); else { R.push_back(v); level[v] = bestk + 1; }
            }
            S.swap(S2);
        }
        if (S.empty() || L.empty() || R.empty()) {
            // (an empty separator: the halves are different components -- keep one vertex as a trivial separator so that
            // the tree stays binary)
            if (L.empty() || R.empty()) return new_front(verts, -1, -1);
            S.push_back(L.back()); L.pop_back();
            if (L.empty()) return new_front(verts, -1, -1);
        }
        std::vector<int>().swap(verts);                          // the recursion keeps only what it needs
        const int c0 = dissect(L);
        const int c1 = dissect(R);
        return new_front(S, c0, c1);
    }
};

// rowptr / col: the caller's CSR pattern (any order inside a row, diagonal optional).  b_identity: B = I, so every diagonal
// position is assembled whether or not A stores it.  leaf: largest subset eliminated as one leaf front.
static inline int make_plan(int N, const std::vector<int>& rowptr, const std::vector<int>& col, bool b_identity, int leaf, plan& P) {
    P = plan();
    P.N = N;
    if (N <= 0 || (int)rowptr.size() != N + 1) return 1;
    builder b;
    b.N = N; b.P = &P; b.leaf = std::max(8, leaf);
    {   // pattern U pattern^T without the diagonal, duplicates removed
        std::vector<int> deg(N, 0);
        for (int i = 0; i < N; ++i)
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                const int j = col[k];
                if (j < 0 || j >= N) return 1;
                if (j != i) { deg[i]++; deg[j]++; }
            }
        std::vector<int> xa(N + 1, 0);
        for (int i = 0; i < N; ++i) xa[i + 1] = xa[i] + deg[i];
        std::vector<int> ad(xa[N]), fill(xa.begin(), xa.end() - 1);
        for (int i = 0; i < N; ++i)
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                const int j = col[k];
                if (j != i) { ad[fill[i]++] = j; ad[fill[j]++] = i; }
            }
        b.xadj.assign(N + 1, 0);
        for (int i = 0; i < N; ++i) {
            std::sort(ad.begin() + xa[i], ad.begin() + xa[i + 1]);
            const int m = (int)(std::unique(ad.begin() + xa[i], ad.begin() + xa[i + 1]) - (ad.begin() + xa[i]));
            b.xadj[i + 1] = b.xadj[i] + m;
        }
        b.adj.resize(b.xadj[N]);
        for (int i = 0; i < N; ++i) std::copy(ad.begin() + xa[i], ad.begin() + xa[i] + (b.xadj[i + 1] - b.xadj[i]), b.adj.begin() + b.xadj[i]);
    }
    b.mark.assign(N, 0); b.mark2.assign(N, 0); b.level.assign(N, 0);
    P.perm.assign(N, -1); P.iperm.assign(N, -1);
    {
        std::vector<int> all(N);
        std::iota(all.begin(), all.end(), 0);
        b.dissect(all);
    }
    if (b.next_piv != N) return 2;
    const int nf = (int)P.fronts.size();

    // ---- struct(f), bottom-up (post order: children first)
    {
        std::vector<std::vector<int>> st(nf);
        std::vector<int> tmp;
        for (int f = 0; f < nf; ++f) {
            front& F = P.fronts[f];
            const int last = F.piv0 + F.npiv;                    // indices >= last are outside the front's pivots
            tmp.clear();
            for (int p = F.piv0; p < last; ++p) {
                const int v = P.perm[p];
                for (int k = b.xadj[v]; k < b.xadj[v + 1]; ++k) { const int u = P.iperm[b.adj[k]]; if (u >= last) tmp.push_back(u); }
            }
            for (int c : F.child) if (c >= 0) { for (int u : st[c]) if (u >= last) tmp.push_back(u); else if (u < F.piv0) return 3; std::vector<int>().swap(st[c]); }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            st[f] = tmp;
            F.nbnd = (int)tmp.size();
            F.bnd_off = P.bnd.size();
            P.bnd.insert(P.bnd.end(), tmp.begin(), tmp.end());
            if (F.parent < 0 && F.nbnd != 0) return 3;
        }
    }

    // ---- groups: equal height, similar size
    {
        int maxh = 0;
        for (const front& F : P.fronts) maxh = std::max(maxh, F.height);
        std::vector<std::vector<int>> byh(maxh + 1);
        for (int f = 0; f < nf; ++f) byh[P.fronts[f].height].push_back(f);
        for (int hgt = 0; hgt <= maxh; ++hgt) {
            std::vector<int>& fs = byh[hgt];
            // by pivot class (the padded pivot count sets the panel steps AND pads both factor blocks), then by boundary size
            std::sort(fs.begin(), fs.end(), [&](int a, int c) {
                const int pa = round_up(P.fronts[a].npiv, 32), pc = round_up(P.fronts[c].npiv, 32);
                if (pa != pc) return pa > pc;
                if (P.fronts[a].nbnd != P.fronts[c].nbnd) return P.fronts[a].nbnd > P.fronts[c].nbnd;
                return a < c;
            });
            size_t i = 0;
            while (i < fs.size()) {
                group G;
                G.height = hgt;
                int npm = 0, nbm = 0;
                double exact = 0.0;
                size_t j = i;
                // A group costs (panel steps) x (latency of one step: a chain of small launches, ~0.2 ms whatever the batch)
                // + (padded flops) / (MFMA rate); in flops per quadrature node one step is worth about step_flops.  A front
                // joins while that is cheaper than a group of its own.
                // The factor store keeps a group's padded geometry, so a second rule bounds the memory: a front does not join
                // when the group's padded store would exceed store_slack x what its members need on their own.
                const double step_flops = 2.5e8;
                const double store_slack = plan_store_slack();
                auto store_of = [](int np_, int nb_) { return (double)(np_ + nb_) * np_ + 64.0 * np_ + (double)np_ * nb_; };
                double cost = 0.0, own_store = 0.0;
                for (; j < fs.size(); ++j) {
                    const front& F = P.fronts[fs[j]];
                    const int np2 = round_up(std::max(npm, F.npiv), 32), nb2 = round_up(std::max(nbm, F.nbnd), 16);
                    const double merged = (np2 / 32) * step_flops + (double)(j - i + 1) * partial_lu_flops(np2 + nb2, np2);
                    const int np1 = round_up(F.npiv, 32), nb1 = round_up(F.nbnd, 16);
                    const double alone = (np1 / 32) * step_flops + partial_lu_flops(np1 + nb1, np1);
                    if (j > i && merged > cost + alone) break;
                    if (j > i && (double)(j - i + 1) * store_of(np2, nb2) > store_slack * (own_store + store_of(np1, nb1))) break;
                    if (j - i >= 2048) break;                    // (fronts x quadrature nodes is a grid dimension)
                    npm = std::max(npm, F.npiv); nbm = std::max(nbm, F.nbnd);
                    exact += partial_lu_flops(F.npiv + F.nbnd, F.npiv);
                    own_store += store_of(np1, nb1);
                    cost = merged;
                }
                G.np = round_up(npm, 32); G.nb = round_up(nbm, 16); G.n = G.np + G.nb;
                G.fronts.assign(fs.begin() + i, fs.begin() + j);
                G.inv128 = G.np >= 256 ? 1 : 0;
                G.flops = (double)G.fronts.size() * partial_lu_flops(G.n, G.np);
                P.flops += G.flops; P.flops_exact += exact;
                const int gid = (int)P.groups.size();
                for (size_t q = 0; q < G.fronts.size(); ++q) { P.fronts[G.fronts[q]].group = gid; P.fronts[G.fronts[q]].slot = (int)q; }
                P.groups.push_back(G);
                i = j;
            }
        }
    }

    // ---- storage offsets.  Work arena: a group's fronts live from its assembly to the assembly of the last parent group;
    // first-fit placement over those lifetimes (groups are processed in index order).
    {
        // (the groups of one height may run side by side on several streams: a region is taken when its LEVEL starts and
        //  given back when the level of its last parent group has ended)
        const int ng = (int)P.groups.size();
        std::vector<int> lev_first(ng), lev_last(ng);
        for (int g = 0; g < ng;) {
            int e = g;
            while (e + 1 < ng && P.groups[e + 1].height == P.groups[g].height) ++e;
            for (int q = g; q <= e; ++q) { lev_first[q] = g; lev_last[q] = e; }
            g = e + 1;
        }
        std::vector<int> last_use(ng);
        for (int g = 0; g < ng; ++g) {
            last_use[g] = g;
            for (int f : P.groups[g].fronts) { const int p = P.fronts[f].parent; if (p >= 0) last_use[g] = std::max(last_use[g], P.fronts[p].group); }
            last_use[g] = lev_last[last_use[g]];
        }
        struct placed { size_t off, len; int last; };
        std::vector<placed> live;
        for (int g = 0; g < ng; ++g) {
            group& G = P.groups[g];
            G.work_per = (size_t)G.n * G.n + inv32_elems(G.np);
            const size_t len = G.work_per * G.fronts.size();
            live.erase(std::remove_if(live.begin(), live.end(), [&](const placed& p) { return p.last < lev_first[g]; }), live.end());
            std::sort(live.begin(), live.end(), [](const placed& a, const placed& c) { return a.off < c.off; });
            size_t off = 0;
            for (const placed& p : live) { if (off + len <= p.off) break; off = std::max(off, p.off + p.len); }
            G.work_off = off;
            live.push_back({off, len, last_use[g]});
            P.work_elems = std::max(P.work_elems, off + len);
            G.store_per = (size_t)G.n * G.np + inv32_elems(G.np) + (G.inv128 ? inv128_elems(G.np) : 0) + (size_t)G.np * G.nb;
            G.store_off = P.store_elems; P.store_elems += G.store_per * G.fronts.size();
            G.piv_off = P.piv_ints; P.piv_ints += (size_t)2 * G.np * G.fronts.size();
            G.rhs_off = P.rhs_rows; P.rhs_rows += (size_t)G.n * G.fronts.size();
            P.max_n = std::max(P.max_n, G.n); P.max_np = std::max(P.max_np, G.np);
            if (G.work_per * G.fronts.size() > (size_t)0x7fffffff) return 4;     // assembly offsets are 32-bit
        }
    }

    // ---- assembly list (every CSR entry goes to the front that owns the earlier-eliminated of its two indices) and
    // extend-add maps, front by front over a scratch map new index -> row of the current front
    {
        std::vector<int> owner(N);                               // new index -> front
        for (int f = 0; f < nf; ++f) for (int p = 0; p < P.fronts[f].npiv; ++p) owner[P.fronts[f].piv0 + p] = f;
        const size_t nnz = (size_t)rowptr[N];
        // entries bucketed by owning front (counting sort), B = I diagonals without an A entry appended per front
        std::vector<int> erow(nnz), efront(nnz);
        std::vector<size_t> fstart(nf + 1, 0);
        std::vector<char> has_diag(b_identity ? N : 0, 0);
        for (int i = 0; i < N; ++i) {
            const int ni = P.iperm[i];
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                const int f = owner[std::min(ni, P.iperm[col[k]])];
                erow[k] = i; efront[k] = f; fstart[f + 1]++;
                if (b_identity && col[k] == i) has_diag[i] = 1;
            }
        }
        for (int f = 0; f < nf; ++f) fstart[f + 1] += fstart[f];
        std::vector<int> byfront(nnz);
        {
            std::vector<size_t> fill(fstart.begin(), fstart.end() - 1);
            for (size_t k = 0; k < nnz; ++k) byfront[fill[efront[k]]++] = (int)k;
        }
        const int ngr = (int)P.groups.size();
        std::vector<int> e_dst, e_src, e_grp;
        e_dst.reserve(nnz + (b_identity ? N : 0)); e_src.reserve(e_dst.capacity()); e_grp.reserve(e_dst.capacity());
        std::vector<int> pos(N, -1);
        P.rel.assign(P.bnd.size(), -1);
        for (int f = 0; f < nf; ++f) {
            const front& F = P.fronts[f];
            const group& G = P.groups[F.group];
            for (int p = 0; p < F.npiv; ++p) pos[F.piv0 + p] = p;
            for (int q = 0; q < F.nbnd; ++q) pos[P.bnd[F.bnd_off + q]] = G.np + q;
            const size_t base = (size_t)F.slot * G.work_per;
            for (size_t e = fstart[f]; e < fstart[f + 1]; ++e) {
                const int k = byfront[e], i = erow[k], j = col[k];
                const int r = pos[P.iperm[i]], c = pos[P.iperm[j]];
                if (r < 0 || c < 0) return 5;
                const int dst = (int)(base + (size_t)r + (size_t)c * G.n);
                e_dst.push_back(b_identity && i == j ? ~dst : dst); e_src.push_back(k); e_grp.push_back(F.group);
            }
            if (b_identity)
                for (int p = 0; p < F.npiv; ++p) if (!has_diag[P.perm[F.piv0 + p]]) {
                    e_dst.push_back(~(int)(base + (size_t)p + (size_t)p * G.n)); e_src.push_back(-1); e_grp.push_back(F.group);
                }
            for (int c : F.child) if (c >= 0) {
                const front& C = P.fronts[c];
                for (int q = 0; q < C.nbnd; ++q) {
                    const int r = pos[P.bnd[C.bnd_off + q]];
                    if (r < 0) return 6;
                    P.rel[C.bnd_off + q] = r;
                }
                P.groups[F.group].kids[F.child[0] == c ? 0 : 1].push_back(c);
            }
            for (int p = 0; p < F.npiv; ++p) pos[F.piv0 + p] = -1;
            for (int q = 0; q < F.nbnd; ++q) pos[P.bnd[F.bnd_off + q]] = -1;
        }
        // counting sort by group: the list is read group by group on the device
        std::vector<size_t> start(ngr + 1, 0);
        for (int g : e_grp) start[g + 1]++;
        for (int g = 0; g < ngr; ++g) start[g + 1] += start[g];
        P.asm_dst.resize(e_dst.size()); P.asm_src.resize(e_dst.size());
        for (int g = 0; g < ngr; ++g) { P.groups[g].asm_begin = start[g]; P.groups[g].asm_end = start[g + 1]; }
        std::vector<size_t> fill(start.begin(), start.end() - 1);
        for (size_t q = 0; q < e_dst.size(); ++q) { const size_t at = fill[e_grp[q]]++; P.asm_dst[at] = e_dst[q]; P.asm_src[at] = e_src[q]; }
    }
    return 0;
}

}   // namespace fh_mf
