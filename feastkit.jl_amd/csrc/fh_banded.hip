// Batched banded LU (ZGBTRF / ZGBTRS semantics) for CSR input with a narrow band: the direct solver
// of the reference's banded drivers (LAPACK.gbtrf!/gbtrs!, src/banded/feast_banded.jl:100-150) and
// a sparse DIRECT path for band matrices (the reference's sparse default is UMFPACK,
// src/sparse/feast_sparse.jl:339, which is not replicated).
//
// Storage per quadrature node: LAPACK general band storage AB(ldab, N), ldab = 2 kl + ku + 1,
//   A(i, j) = AB[kv + i - j + j * ldab],  kv = kl + ku,   max(0, j - ku) <= i <= min(N - 1, j + kl),
// the first kl rows are the fill-in space of the row interchanges; complex128, 0-based here.
// One workgroup per node walks the columns (the elimination is inherently sequential in j; the
// parallelism is nodes x the (kl x (kl+ku)) window, and nodes x right-hand sides in the solves).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "fh_banded.hpp"
#include "fh_dense.hpp"
#include <string>
static inline cplx fh_ing_zero(cplx) { return cmake(0, 0); }
static inline cplx fh_ing_add(cplx a, cplx b) { return cadd(a, b); }
#define FH_INGEST_STORAGE_CSR 0
#include "fh_ingest.hpp"       // fh_rcm, fh_bandwidth
#include "../../include/feasthip.h"

#define FH_BLOCK 256
#define BAND_THREADS 512

// AB = z B - A from the CSR arrays (A and B share the union pattern); AB was zeroed by the caller
template <typename VT, bool BIDENT>
__global__ __launch_bounds__(FH_BLOCK) void k_band_form(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                         const VT* __restrict__ aval, const VT* __restrict__ bval,
                                                         cplx* const* ABs, const cplx* z, int N, int kl, int ku) {
    cplx* AB = ABs[blockIdx.y];
    const cplx zz = z[blockIdx.y];
    const int ldab = 2 * kl + ku + 1, kv = kl + ku;
    const int i = blockIdx.x * FH_BLOCK + threadIdx.x;
    if (i >= N) return;
    if (BIDENT) AB[kv + (size_t)i * ldab] = zz;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int j = col[k];
        cplx a;
        if constexpr (sizeof(VT) == sizeof(cplx)) a = cmake(aval[k].x, aval[k].y); else a = cmake(aval[k], 0.0);
        cplx* dst = AB + (size_t)j * ldab + kv + i - j;
        if (BIDENT) {
            *dst = csub(*dst, a);
        } else {
            cplx b;
            if constexpr (sizeof(VT) == sizeof(cplx)) b = cmake(bval[k].x, bval[k].y); else b = cmake(bval[k], 0.0);
            *dst = csub(cmul(zz, b), a);
        }
    }
}

// Unblocked band LU with partial pivoting (ZGBTF2): pivot rule IZAMAX (max |re|+|im|, lowest index on
// ties) among the kl+1 candidates of the column; the rank-1 update always spans the full kl+ku
// columns to the right (LAPACK trims it to `ju`; the extra entries are zeros).
__global__ __launch_bounds__(BAND_THREADS) void k_band_lu(cplx* const* ABs, int* const* pivs, int N, int kl, int ku,
                                                          int* info) {
    cplx* AB = ABs[blockIdx.x];
    int* ipiv = pivs[blockIdx.x];
    const int ldab = 2 * kl + ku + 1, kv = kl + ku;
    const int t = threadIdx.x;
    extern __shared__ cplx sm[];
    cplx* lmul = sm;                 // [kl]      multipliers of the current column
    cplx* urow = sm + kl;            // [kv + 1]  pivot row, columns j .. j + kv
    __shared__ int s_jp;
    for (int j = 0; j < N; ++j) {
        const int km = min(kl, N - 1 - j);
        const int nc = min(kv, N - 1 - j) + 1;
        cplx* colj = AB + (size_t)j * ldab + kv;          // A(j + r, j) = colj[r]
        if (t < 64) {
            double best = -1.0;
            int br = 0x7fffffff;
            for (int r = t; r <= km; r += 64) {
                const cplx v = colj[r];
                const double m = fabs(v.x) + fabs(v.y);
                if (m > best) { best = m; br = r; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_xor(best, off);
                const int orr = __shfl_xor(br, off);
                if (o > best || (o == best && orr < br)) { best = o; br = orr; }
            }
            if (t == 0) {
                s_jp = br == 0x7fffffff ? 0 : br;
                ipiv[j] = j + s_jp;
                if ((!(best > 0.0) || !isfinite(best)) && info[blockIdx.x] == 0) info[blockIdx.x] = j + 1;
            }
        }
        __syncthreads();
        const int jp = s_jp;
        // row interchange over columns j .. j + nc - 1, pivot row kept in LDS
        for (int cc = t; cc < nc; cc += BAND_THREADS) {
            cplx* pc = AB + (size_t)(j + cc) * ldab + kv - cc;     // A(j, j + cc)
            const cplx top = pc[0], piv = pc[jp];
            if (jp != 0) { pc[0] = piv; pc[jp] = top; }
            urow[cc] = piv;
        }
        __syncthreads();
        const cplx pv = urow[0];
        const bool singular = (pv.x == 0.0 && pv.y == 0.0);
        const cplx inv = singular ? cmake(0, 0) : cdiv(cmake(1, 0), pv);
        for (int r = 1 + t; r <= km; r += BAND_THREADS) {
            const cplx l = cmul(colj[r], inv);
            colj[r] = l;
            lmul[r - 1] = l;
        }
        __syncthreads();
        // A(j + r, j + cc) -= l_r * u_cc
        const int work = km * (nc - 1);
        for (int e = t; e < work; e += BAND_THREADS) {
            const int r = 1 + e % km, cc = 1 + e / km;
            cplx* p = AB + (size_t)(j + cc) * ldab + kv - cc + r;
            *p = csub(*p, cmul(lmul[r - 1], urow[cc]));
        }
        __syncthreads();
    }
}

// Y[node] = RHS (shared right-hand side panel, row-major N x ld)
__global__ __launch_bounds__(FH_BLOCK) void k_band_copy_rhs(const cplx* __restrict__ RHS, size_t rhs_stride, cplx* __restrict__ Y, size_t stride,
                                                             size_t total) {
    cplx* Yn = Y + (size_t)blockIdx.y * stride;
    const cplx* Rn = RHS + (size_t)blockIdx.y * rhs_stride;     // rhs_stride = 0: one right-hand side panel shared by all nodes
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) Yn[e] = Rn[e];
}

// ZGBTRS on a row-major N x ld panel: one workgroup per (16-column tile, node); thread = (row lane, column)
__global__ __launch_bounds__(FH_BLOCK) void k_band_solve(cplx* const* ABs, int* const* pivs, cplx* Y, size_t stride, int N,
                                                          int ld, int kl, int ku) {
    const cplx* AB = ABs[blockIdx.y];
    const int* ipiv = pivs[blockIdx.y];
    cplx* Yn = Y + (size_t)blockIdx.y * stride + 16 * blockIdx.x;
    const int ldab = 2 * kl + ku + 1, kv = kl + ku;
    const int c = threadIdx.x & 15, rr = threadIdx.x >> 4;
    constexpr int RL = FH_BLOCK / 16;
    // forward: L y = P b, interchanges applied on the fly
    for (int j = 0; j < N; ++j) {
        const int km = min(kl, N - 1 - j);
        const int p = ipiv[j];
        if (p != j && rr == 0) {
            const cplx u = Yn[(size_t)j * ld + c];
            Yn[(size_t)j * ld + c] = Yn[(size_t)p * ld + c];
            Yn[(size_t)p * ld + c] = u;
        }
        __syncthreads();
        const cplx yj = Yn[(size_t)j * ld + c];
        const cplx* colj = AB + (size_t)j * ldab + kv;
        for (int r = 1 + rr; r <= km; r += RL) {
            cplx* y = Yn + (size_t)(j + r) * ld + c;
            *y = csub(*y, cmul(colj[r], yj));
        }
        __syncthreads();
    }
    // backward: U x = y, U has kv super-diagonals
    for (int j = N - 1; j >= 0; --j) {
        const cplx* colj = AB + (size_t)j * ldab + kv;
        if (rr == 0) Yn[(size_t)j * ld + c] = cdiv(Yn[(size_t)j * ld + c], colj[0]);
        __syncthreads();
        const cplx yj = Yn[(size_t)j * ld + c];
        const int nr = min(kv, j);
        for (int r = 1 + rr; r <= nr; r += RL) {
            cplx* y = Yn + (size_t)(j - r) * ld + c;
            *y = csub(*y, cmul(colj[-r], yj));
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// host orchestration
//
// Band plan (made once per matrix, on first use).  The pattern is measured in the order it is stored on the device (the
// ingest may have renumbered it).  A band of at most FH_BAND_NARROW (kl + ku <= 64) in that order goes to the one-workgroup
// elimination above: it walks the columns one by one (two barriers each, the window through one CU), which is fine for a
// few diagonals and hopeless for hundreds -- a 250 x 200 grid (kl = ku = 250, N = 50 000) took 6.6 s there against 0.5 s
// in the blocked solver.  Anything wider is renumbered by reverse Cuthill-McKee (fh_ingest.hpp) if that narrows it and goes to
// the blocked band LU on the dense kernels (fh_dense.hip, fh_wband_*): the sparse direct solver for general patterns.
// FH_WBAND=1 sends every matrix there (tests).
// ---------------------------------------------------------------------------------------
#define FH_BAND_NARROW 64

void fh_banded_free(feasthip_ctx* h) {
    for (void* p : h->band_factors) if (p) hipFree(p);
    for (int* p : h->band_pivots) if (p) hipFree(p);
    h->band_factors.clear(); h->band_pivots.clear(); h->band_valid.clear(); h->band_z.clear();
    if (h->band_perm) hipFree(h->band_perm);
    if (h->band_iperm) hipFree(h->band_iperm);
    h->band_perm = nullptr; h->band_iperm = nullptr;
    h->band_plan = 0; h->band_kl = 0; h->band_ku = 0;
    fh_mf_free(h);
}

static int band_make_plan(feasthip_ctx* h) {
    if (h->band_plan) return 0;
    if (h->kind != 2) { h->last_error = "banded LU needs a CSR matrix (feasthip_set_csr)"; return FEASTHIP_ERROR_FPM; }
    const int64_t N = h->csr.N;
    if ((int64_t)h->host_rowptr.size() != N + 1) { h->last_error = "banded LU: no host pattern"; return FEASTHIP_ERROR_INTERNAL; }
    int kl0 = 0, ku0 = 0;
    fh_bandwidth(N, h->host_rowptr, h->host_col, nullptr, kl0, ku0);
    const bool force_wide = getenv("FH_WBAND") && atoi(getenv("FH_WBAND")) != 0;     // read per plan: tests switch it (and it keeps the multifrontal plan out)
    const int mf_mode = getenv("FH_MF") ? atoi(getenv("FH_MF")) : -1;                 // 0 never, 1 always (tests), else by predicted work
    if (!force_wide && mf_mode != 1 && kl0 + ku0 <= FH_BAND_NARROW) {
        h->band_plan = 1; h->band_kl = kl0; h->band_ku = ku0;
        return 0;
    }
    std::vector<int> perm, iperm(N);
    fh_rcm(N, h->host_rowptr, h->host_col, perm);
    for (int64_t i = 0; i < N; ++i) iperm[perm[i]] = (int)i;
    int kl1 = 0, ku1 = 0;
    fh_bandwidth(N, h->host_rowptr, h->host_col, iperm.data(), kl1, ku1);
    // the elimination costs N kl (kl + ku): compare that, not the plain width
    if ((double)kl1 * (kl1 + ku1) >= (double)kl0 * (kl0 + ku0)) {
        for (int64_t i = 0; i < N; ++i) { perm[i] = (int)i; iperm[i] = (int)i; }
        kl1 = kl0; ku1 = ku0;
    }
    // Multifrontal plan (fh_mf.hpp, numeric phase in fh_dense.hip): fill confined to the fronts of a nested-dissection tree
    // instead of the band.  Taken when its (padded) work is under half the band elimination's -- the band LU streams one
    // long trailing update per block column, the fronts are many smaller batches.
    // FH_MF=0 never, FH_MF=1 always (tests); FH_MF_LEAF: largest leaf subset (default 64).
    {
        const int mode = mf_mode;
        if (mode != 0 && !(force_wide && mode != 1) && (N >= 2048 || mode == 1)) {
            const int leaf = getenv("FH_MF_LEAF") ? std::max(8, atoi(getenv("FH_MF_LEAF"))) : 64;
            const auto t0 = std::chrono::steady_clock::now();
            if (fh_mf_make_plan(h, leaf) == 0) {
                const double band_flops = 8.0 * (double)N * (double)kl1 * (double)(kl1 + ku1);
                // (a front beyond 16 384 rows is outside the panel kernels' reach: such a pattern has no small separators anyway)
                const bool take = fh_mf_max_front(h) <= 16384 && (mode == 1 || fh_mf_plan_flops(h) < 0.5 * band_flops);
                if (getenv("FH_DEBUG_TIMING"))
                    fprintf(stderr, "[feasthip] multifrontal plan (%.1f ms): %.3e flop and %.2f GB per node, band %.3e flop and %.2f GB -> %s\n",
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), fh_mf_plan_flops(h),
                            fh_mf_store_bytes(h, 64) / 1e9, band_flops, (double)fh_wband_elems((int)N, kl1, ku1) * sizeof(cplx) / 1e9, take ? "multifrontal" : "band");
                if (take) { h->band_plan = 3; h->band_kl = kl1; h->band_ku = ku1; return 0; }
                fh_mf_free(h);
            } else {
                h->last_error.clear();
            }
        }
    }
    FH_CHECK(hipMalloc((void**)&h->band_perm, N * sizeof(int)));
    FH_CHECK(hipMalloc((void**)&h->band_iperm, N * sizeof(int)));
    FH_CHECK(hipMemcpy(h->band_perm, perm.data(), N * sizeof(int), hipMemcpyHostToDevice));
    FH_CHECK(hipMemcpy(h->band_iperm, iperm.data(), N * sizeof(int), hipMemcpyHostToDevice));
    h->band_plan = 2; h->band_kl = kl1; h->band_ku = ku1;
    if (getenv("FH_DEBUG_TIMING"))
        fprintf(stderr, "[feasthip] band plan: stored order kl %d ku %d, band order kl %d ku %d, %.2f GB per node\n", kl0, ku0, kl1, ku1,
                (double)fh_wband_elems((int)N, kl1, ku1) * sizeof(cplx) / 1e9);
    return 0;
}

static size_t band_slot_bytes(feasthip_ctx* h) {
    const size_t N = (size_t)h->csr.N;
    if (h->band_plan == 3) return fh_mf_store_bytes(h, h->band_prec);
    if (h->band_plan == 2) return fh_wband_elems((int)N, h->band_kl, h->band_ku) * (h->band_prec == 32 ? sizeof(cplxf) : sizeof(cplx));
    return ((size_t)2 * h->band_kl + h->band_ku + 1) * N * sizeof(cplx);
}

static int band_check(feasthip_ctx* h) {
    int rc = band_make_plan(h);
    if (rc) return rc;
    if (h->band_plan == 1) {
        const size_t lds = (size_t)(2 * h->band_kl + h->band_ku + 1) * sizeof(cplx);
        if (lds > 60000) { h->last_error = "banded LU: internal (narrow plan with a wide band)"; return FEASTHIP_ERROR_INTERNAL; }
    } else if (h->band_plan == 2 && h->band_kl > 12000) {
        h->last_error = "banded LU: band too wide after reordering (kl > 12000); use an iterative solver";
        return FEASTHIP_ERROR_FPM;
    }
    return 0;
}

static int band_ensure_slots(feasthip_ctx* h, int nslots) {
    const size_t N = (size_t)h->csr.N;
    // complex64 factors (feasthip_set_solver factor_precision = 32) exist for the blocked plan only; the caller refines in fp64
    const int prec = (h->band_plan >= 2 && h->factor_precision == 32) ? 32 : 64;
    if (prec != h->band_prec) {
        for (void* p : h->band_factors) if (p) hipFree(p);
        for (int* p : h->band_pivots) if (p) hipFree(p);
        h->band_factors.clear(); h->band_pivots.clear(); h->band_valid.clear(); h->band_z.clear();
        h->band_prec = prec;
    }
    const size_t bytes = band_slot_bytes(h);
    const int missing = nslots - (int)h->band_factors.size();
    if (missing > 0) {
        size_t free_b = 0, total_b = 0;
        // (multifrontal plan: the work arena and the substitution panels are transient buffers of about the factors' size again)
        const double transient = h->band_plan == 3 ? (double)nslots * ((double)fh_mf_work_bytes(h, prec) + 0.5 * (double)bytes) : 0.0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)missing * (double)bytes + transient > 0.92 * (double)free_b) {
            h->last_error = "banded LU: " + std::to_string(missing) + " factors of " + std::to_string(bytes >> 20) + " MiB do not fit the free device memory (" +
                            std::to_string(free_b >> 20) + " MiB)";
            return FEASTHIP_ERROR_MEMORY;
        }
    }
    const auto t_alloc = std::chrono::steady_clock::now();
    const int had = (int)h->band_factors.size();
    struct alloc_report {
        feasthip_ctx* h; std::chrono::steady_clock::time_point t0; int had; size_t bytes;
        ~alloc_report() {
            if ((int)h->band_factors.size() > had && getenv("FH_DEBUG_TIMING"))
                fprintf(stderr, "[feasthip] band factors: %d x %.2f GB allocated in %.1f ms\n", (int)h->band_factors.size() - had, bytes / 1e9,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    } report{h, t_alloc, had, bytes};
    while ((int)h->band_factors.size() < nslots) {
        void* f = nullptr; int* pv = nullptr;
        if (hipMalloc(&f, bytes) != hipSuccess) { (void)hipGetLastError(); h->last_error = "hipMalloc(band factor)"; return FEASTHIP_ERROR_MEMORY; }
        if (hipMalloc((void**)&pv, (h->band_plan == 3 ? fh_mf_pivot_ints(h) : N) * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); hipFree(f); h->last_error = "hipMalloc(band pivots)"; return FEASTHIP_ERROR_MEMORY; }
        h->band_factors.push_back(f); h->band_pivots.push_back(pv); h->band_valid.push_back(0); h->band_z.push_back(cmake(0, 0));
    }
    return 0;
}

// device arrays of per-node pointers for the slots in `which`: storage (narrow) or matrix base (wide), pivots, (wide) perm
static int band_pointer_arrays(feasthip_ctx* h, const std::vector<int>& which, cplx*** dabs_out, int*** dpvs_out, int*** dperms_out) {
    const int nf = (int)which.size();
    const size_t off = h->band_plan == 2 ? fh_wband_base_offset((int)h->csr.N, h->band_kl, h->band_ku) : 0;
    void* p;
    int rc;
    std::vector<cplx*> abs(nf);
    std::vector<int*> pvs(nf), perms(nf, h->band_perm);
    const size_t esz = (h->band_plan == 2 && h->band_prec == 32) ? sizeof(cplxf) : sizeof(cplx);
    for (int q = 0; q < nf; ++q) { abs[q] = (cplx*)((char*)h->band_factors[which[q]] + off * esz); pvs[q] = h->band_pivots[which[q]]; }
    if ((rc = fh_get_buf(h, "bd_ptrs", nf * sizeof(cplx*), &p))) return rc;
    cplx** dabs = (cplx**)p;
    if ((rc = fh_get_buf(h, "bd_pptrs", nf * sizeof(int*), &p))) return rc;
    int** dpvs = (int**)p;
    if ((rc = fh_get_buf(h, "bd_permptrs", nf * sizeof(int*), &p))) return rc;
    int** dperms = (int**)p;
    FH_CHECK(hipMemcpyAsync(dabs, abs.data(), nf * sizeof(cplx*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(dpvs, pvs.data(), nf * sizeof(int*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(dperms, perms.data(), nf * sizeof(int*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));      // the host vectors go out of scope
    *dabs_out = dabs; *dpvs_out = dpvs; *dperms_out = dperms;
    return 0;
}

// (fronts of a group) x (nodes of a call) is a grid dimension of the multifrontal kernels (FH_MF_NODES_PER_CALL: smaller batches, tests)
static int mf_nodes_per_call(feasthip_ctx* h) {
    int per = std::max(1, 65535 / std::max(1, fh_mf_max_group(h)));
    if (getenv("FH_MF_NODES_PER_CALL")) per = std::max(1, std::min(per, atoi(getenv("FH_MF_NODES_PER_CALL"))));
    return per;
}

static int band_factor_batch(feasthip_ctx* h, const std::vector<int>& which, const std::vector<cplx>& zlist, std::vector<int>& info_out) {
    const int nf = (int)which.size();
    info_out.assign(nf, 0);
    if (nf == 0) return 0;
    const int N = (int)h->csr.N, kl = h->band_kl, ku = h->band_ku;
    const auto t_factor = std::chrono::steady_clock::now();
    void* p;
    int rc;
    if (h->band_plan == 3) {
        std::vector<void*> stores(nf);
        std::vector<int*> pvs(nf);
        for (int q = 0; q < nf; ++q) { stores[q] = h->band_factors[which[q]]; pvs[q] = h->band_pivots[which[q]]; }
        if ((rc = fh_get_buf(h, "bd_z", nf * sizeof(cplx), &p))) return rc;
        cplx* dz = (cplx*)p;
        FH_CHECK(hipMemcpyAsync(dz, zlist.data(), nf * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
        // (fronts of a group) x (nodes of a call) is a grid dimension: node batches for long contours
        const int per_call = mf_nodes_per_call(h);
        for (int q0 = 0; q0 < nf; q0 += per_call) {
            const int cnt = std::min(per_call, nf - q0);
            std::vector<int> info_part;
            if ((rc = fh_mf_factor(h, h->band_prec, cnt, stores.data() + q0, pvs.data() + q0, dz + q0, info_part))) return rc;
            for (int q = 0; q < cnt; ++q) info_out[q0 + q] = info_part[q];
        }
        if (getenv("FH_DEBUG_TIMING"))
            fprintf(stderr, "[feasthip] multifrontal LU: %d factorisations (%d-bit) in %.1f ms\n", nf, h->band_prec,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_factor).count());
        return 0;
    }
    cplx** dabs; int** dpvs; int** dperms;
    if ((rc = band_pointer_arrays(h, which, &dabs, &dpvs, &dperms))) return rc;
    if ((rc = fh_get_buf(h, "bd_z", nf * sizeof(cplx), &p))) return rc;
    cplx* dz = (cplx*)p;
    if ((rc = fh_get_buf(h, "bd_info", nf * sizeof(int), &p))) return rc;
    int* dinfo = (int*)p;
    FH_CHECK(hipMemcpyAsync(dz, zlist.data(), nf * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemsetAsync(dinfo, 0, nf * sizeof(int), h->stream));
    if (h->band_plan == 2) {
        std::vector<void*> abs(nf);
        for (int q = 0; q < nf; ++q) abs[q] = h->band_factors[which[q]];
        if ((rc = fh_wband_factor(h, h->band_prec, nf, abs.data(), (void**)dabs, dpvs, dz, dinfo, h->band_iperm, kl, ku))) return rc;
    } else {
        const size_t ldab = (size_t)2 * kl + ku + 1;
        for (int q = 0; q < nf; ++q) FH_CHECK(hipMemsetAsync(h->band_factors[which[q]], 0, ldab * N * sizeof(cplx), h->stream));
        const dim3 grid((N + FH_BLOCK - 1) / FH_BLOCK, nf), block(FH_BLOCK);
        const bool bid = h->csr.b_identity != 0;
        fh_prof_begin(h, "band_form");
        if (h->csr.is_complex) {
            if (bid) hipLaunchKernelGGL((k_band_form<cplx, true>), grid, block, 0, h->stream, h->csr.rowptr, h->csr.col, (const cplx*)h->csr.aval, (const cplx*)nullptr, dabs, dz, N, kl, ku);
            else hipLaunchKernelGGL((k_band_form<cplx, false>), grid, block, 0, h->stream, h->csr.rowptr, h->csr.col, (const cplx*)h->csr.aval, (const cplx*)h->csr.bval, dabs, dz, N, kl, ku);
        } else {
            if (bid) hipLaunchKernelGGL((k_band_form<double, true>), grid, block, 0, h->stream, h->csr.rowptr, h->csr.col, (const double*)h->csr.aval, (const double*)nullptr, dabs, dz, N, kl, ku);
            else hipLaunchKernelGGL((k_band_form<double, false>), grid, block, 0, h->stream, h->csr.rowptr, h->csr.col, (const double*)h->csr.aval, (const double*)h->csr.bval, dabs, dz, N, kl, ku);
        }
        fh_prof_end(h);
        fh_prof_begin(h, "band_lu");
        hipLaunchKernelGGL(k_band_lu, dim3(nf), dim3(BAND_THREADS), (size_t)(kl + kl + ku + 1) * sizeof(cplx), h->stream, dabs, dpvs, N, kl, ku, dinfo);
        fh_prof_end(h);
    }
    FH_CHECK(hipMemcpyAsync(info_out.data(), dinfo, nf * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (getenv("FH_DEBUG_TIMING"))
        fprintf(stderr, "[feasthip] band LU: %d factorisations (plan %d, kl %d ku %d, %d-bit) in %.1f ms\n", nf, h->band_plan, kl, ku, h->band_prec,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_factor).count());
    return 0;
}

static int band_solve_batch(feasthip_ctx* h, int ld, int m, const std::vector<int>& slots, const cplx* RHS, size_t rhs_stride, cplx* Y, size_t stride) {
    const int nf = (int)slots.size();
    const int N = (int)h->csr.N;
    int rc;
    if (h->band_plan == 3) {
        std::vector<void*> stores(nf);
        std::vector<int*> pvs(nf);
        for (int q = 0; q < nf; ++q) { stores[q] = h->band_factors[slots[q]]; pvs[q] = h->band_pivots[slots[q]]; }
        const int per_call = mf_nodes_per_call(h);
        for (int q0 = 0; q0 < nf; q0 += per_call) {
            const int cnt = std::min(per_call, nf - q0);
            if ((rc = fh_mf_solve(h, h->band_prec, cnt, stores.data() + q0, pvs.data() + q0, RHS + (size_t)q0 * rhs_stride, rhs_stride, Y + (size_t)q0 * stride, stride, ld, m))) return rc;
        }
        return 0;
    }
    cplx** dabs; int** dpvs; int** dperms;
    if ((rc = band_pointer_arrays(h, slots, &dabs, &dpvs, &dperms))) return rc;
    if (h->band_plan == 2) {
        void* p;
        const size_t bstride = (size_t)N * ld;
        const size_t esz = h->band_prec == 32 ? sizeof(cplxf) : sizeof(cplx);
        if ((rc = fh_get_buf(h, "bd_ypanel", (size_t)nf * bstride * esz, &p))) return rc;
        void* Yb = p;
        if ((rc = fh_get_buf(h, "bd_zpanel", (size_t)nf * bstride * esz, &p))) return rc;
        void* Zb = p;
        return fh_wband_solve(h, h->band_prec, nf, (void**)dabs, dpvs, dperms, h->band_perm, RHS, rhs_stride, Y, stride, Yb, Zb, ld, m, h->band_kl, h->band_ku);
    }
    fh_prof_begin(h, "band_solve");
    const size_t total = (size_t)N * ld;
    hipLaunchKernelGGL(k_band_copy_rhs, dim3((unsigned)std::min<size_t>((total + FH_BLOCK - 1) / FH_BLOCK, 2048), nf), dim3(FH_BLOCK), 0, h->stream, RHS, rhs_stride, Y, stride, total);
    hipLaunchKernelGGL(k_band_solve, dim3(ld / 16, nf), dim3(FH_BLOCK), 0, h->stream, dabs, dpvs, Y, stride, N, ld, h->band_kl, h->band_ku);
    fh_prof_end(h);
    return 0;
}

int fh_banded_solve_nodes(feasthip_ctx* h, int ld, int m, int nodes, const std::vector<cplx>& z, const cplx* RHS, size_t rhs_stride, cplx* Y,
                          size_t stride, std::vector<int>& status, int64_t* nfact) {
    int rc = band_check(h);
    if (rc) return rc;
    if ((rc = band_ensure_slots(h, nodes))) return rc;
    std::vector<int> need;
    std::vector<cplx> zl;
    for (int e = 0; e < nodes; ++e) {
        const bool ok = h->cache_factors && h->band_valid[e] == 1 && h->band_z[e].x == z[e].x && h->band_z[e].y == z[e].y;
        if (!ok) { need.push_back(e); zl.push_back(z[e]); h->band_valid[e] = 0; }
    }
    std::vector<int> info;
    if ((rc = band_factor_batch(h, need, zl, info))) return rc;
    for (size_t q = 0; q < need.size(); ++q) {
        h->band_z[need[q]] = zl[q];
        h->band_valid[need[q]] = info[q] == 0 ? 1 : -1;
    }
    if (nfact) *nfact = (int64_t)need.size();
    std::vector<int> slots(nodes);
    for (int e = 0; e < nodes; ++e) slots[e] = e;
    if ((rc = band_solve_batch(h, ld, m, slots, RHS, rhs_stride, Y, stride))) return rc;
    status.assign(nodes, 0);
    for (int e = 0; e < nodes; ++e) if (h->band_valid[e] != 1) status[e] = FEASTHIP_ERROR_LAPACK;
    return 0;
}

int fh_banded_solve_single(feasthip_ctx* h, int ld, int m, cplx z, const cplx* RHS, cplx* Y, int* status, int64_t* nfact) {
    int rc = band_check(h);
    if (rc) return rc;
    int slot = h->node_count;           // a shift equal to a local quadrature node reuses that node's factor
    for (int e = 0; e < h->node_count && e < (int)h->node_ids.size(); ++e) {
        const cplx ze = h->zne[h->node_ids[e]];
        if (ze.x == z.x && ze.y == z.y) { slot = e; break; }
    }
    if ((rc = band_ensure_slots(h, std::max(slot, h->node_count) + 1))) return rc;
    std::vector<int> need(1, slot), info;
    std::vector<cplx> zl(1, z);
    const bool cached = h->cache_factors && h->band_valid[slot] == 1 && h->band_z[slot].x == z.x && h->band_z[slot].y == z.y;
    if (!cached) {
        if ((rc = band_factor_batch(h, need, zl, info))) return rc;
        h->band_z[slot] = z;
        h->band_valid[slot] = info[0] == 0 ? 1 : -1;
        if (nfact) *nfact = 1;
    }
    if ((rc = band_solve_batch(h, ld, m, need, RHS, 0, Y, (size_t)h->csr.N * ld))) return rc;
    *status = h->band_valid[slot] == 1 ? 0 : FEASTHIP_ERROR_LAPACK;
    return 0;
}

// real flops of ONE node's factorisation under the plan in force (band: 8 N kl (kl + ku); multifrontal: the padded fronts)
int fh_banded_plan_flops(feasthip_ctx* h, double* flops) {
    int rc = band_check(h);
    if (rc && !h->band_plan) return rc;
    if (flops) *flops = h->band_plan == 3 ? fh_mf_plan_flops(h) : 8.0 * (double)h->csr.N * (double)h->band_kl * (double)(h->band_kl + h->band_ku);
    return rc;
}

// The band the direct solver would work on, and the device memory one factor takes (feasthip_band_plan)
int fh_banded_plan(feasthip_ctx* h, int* kl, int* ku, int64_t* bytes_per_node, int* blocked) {
    int rc = band_check(h);
    if (rc && !h->band_plan) return rc;
    if (kl) *kl = h->band_kl;
    if (ku) *ku = h->band_ku;
    if (bytes_per_node) *bytes_per_node = (int64_t)band_slot_bytes(h);
    if (blocked) *blocked = h->band_plan == 3 ? 2 : (h->band_plan == 2 ? 1 : 0);      // 2: multifrontal (kl, ku: the band it replaced)
    return rc;
}
