// fh_blockops.hip -- panel (N x ld row-major c128) kernels around the shifted solves:
// layout conversion at the C-ABI boundary, weighted accumulation Q_proj = sum w_e Y_e
// (src/dense/feast_dense.jl:231, src/sparse/feast_sparse.jl:369), tall-skinny Gram products
// Q^H Y on f64 MFMA (src/dense/feast_dense.jl:252-265, src/kernel/feast_kernel.jl:147-148),
// Ritz back-transform X = Q V (src/dense/feast_dense.jl:287-290), column norms/scaling
// (:301-305), and the rank-revealing orthonormalisation that stands in for
// _feast_qr_compress! (src/core/feast_aux.jl:101-131).
#include <stdlib.h>
#include "fh_common.hpp"
#include "fh_kernels.hpp"

#define FH_BLOCK 256
typedef double v4d __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// layout conversion
// ---------------------------------------------------------------------------------------
template <typename ST>
__global__ __launch_bounds__(FH_BLOCK) void k_to_panel(const ST* __restrict__ src, int64_t lds, int N, int m,
                                                        cplx* __restrict__ dst, int ld, const int* __restrict__ perm) {
    __shared__ cplx tile[64][65];
    const int i0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    const int ti = t & 63, tc = t >> 6;
    const int srow = (i0 + ti < N) ? (perm ? perm[i0 + ti] : i0 + ti) : 0;      // caller's row behind panel row i0 + ti
    for (int c = tc; c < ld; c += 4) {
        cplx v = cmake(0, 0);
        if (c < m && i0 + ti < N) {
            if constexpr (sizeof(ST) == sizeof(cplx)) v = ((const cplx*)src)[(size_t)c * lds + srow];
            else v = cmake(((const double*)src)[(size_t)c * lds + srow], 0.0);
        }
        tile[c][ti] = v;
    }
    __syncthreads();
    for (int e = t; e < 64 * ld; e += FH_BLOCK) {
        int r = e / ld, c = e % ld;
        if (i0 + r < N) dst[(size_t)(i0 + r) * ld + c] = tile[c][r];
    }
}

__global__ __launch_bounds__(FH_BLOCK) void k_from_panel(const cplx* __restrict__ src, int ld, int N, int m,
                                                          cplx* __restrict__ dst, int64_t ldd, const int* __restrict__ perm) {
    __shared__ cplx tile[64][65];
    const int i0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    for (int e = t; e < 64 * ld; e += FH_BLOCK) {
        int r = e / ld, c = e % ld;
        tile[c][r] = (i0 + r < N) ? src[(size_t)(i0 + r) * ld + c] : cmake(0, 0);
    }
    __syncthreads();
    const int ti = t & 63, tc = t >> 6;
    const int drow = (i0 + ti < N) ? (perm ? perm[i0 + ti] : i0 + ti) : 0;
    for (int c = tc; c < m; c += 4)
        if (i0 + ti < N) dst[(size_t)c * ldd + drow] = tile[c][ti];
}

void fh_launch_to_panel(const cplx* src, int64_t lds, int N, int m, cplx* dst, int ld, hipStream_t st, const int* perm) {
    hipLaunchKernelGGL((k_to_panel<cplx>), dim3((N + 63) / 64), dim3(FH_BLOCK), 0, st, src, lds, N, m, dst, ld, perm);
}
void fh_launch_to_panel_real(const double* src, int64_t lds, int N, int m, cplx* dst, int ld, hipStream_t st) {
    hipLaunchKernelGGL((k_to_panel<double>), dim3((N + 63) / 64), dim3(FH_BLOCK), 0, st, src, lds, N, m, dst, ld, (const int*)nullptr);
}
void fh_launch_from_panel(const cplx* src, int ld, int N, int m, cplx* dst, int64_t ldd, hipStream_t st, const int* perm) {
    hipLaunchKernelGGL(k_from_panel, dim3((N + 63) / 64), dim3(FH_BLOCK), 0, st, src, ld, N, m, dst, ldd, perm);
}

// ---------------------------------------------------------------------------------------
// resident refinement loop (feasthip_*_resident): column blocks of panels, packing for the per-loop reduce
// ---------------------------------------------------------------------------------------
// dst (N x ldd panel, zero padded) = columns [c0, c0 + w) of src (N x lds panel)
__global__ __launch_bounds__(FH_BLOCK) void k_panel_cols(const cplx* __restrict__ src, int lds, int c0, int w, size_t total,
                                                          cplx* __restrict__ dst, int ldd) {
    const int c = threadIdx.x % ldd;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const size_t row = e / ldd;
        dst[e] = (c < w) ? src[row * lds + c0 + c] : cmake(0, 0);
    }
}
void fh_launch_panel_cols(const cplx* src, int lds, int c0, int w, int N, cplx* dst, int ldd, hipStream_t st) {
    hipLaunchKernelGGL(k_panel_cols, dim3(fh_vec_nblk(N, ldd)), dim3(FH_BLOCK), 0, st, src, lds, c0, w, (size_t)N * ldd, dst, ldd);
}
// pack[row * ldd * (real ? 1 : 2) + ...] : the w columns of src (N x lds panel) go to columns [c0, c0 + w) of an N x ldd
// array of reals (real_only) or interleaved complex values; every other entry of the array is written as zero, so that the
// sum over ranks with disjoint column blocks assembles the full panel
__global__ __launch_bounds__(FH_BLOCK) void k_pack_cols(const cplx* __restrict__ src, int lds, int c0, int w, size_t total,
                                                         double* __restrict__ dst, int ldd, int real_only) {
    const int c = threadIdx.x % ldd;
    const bool mine = c >= c0 && c < c0 + w;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const size_t row = e / ldd;
        const cplx v = mine ? src[row * lds + (c - c0)] : cmake(0, 0);
        if (real_only) dst[e] = v.x;
        else { dst[2 * e] = v.x; dst[2 * e + 1] = v.y; }
    }
}
void fh_launch_pack_cols(const cplx* src, int lds, int c0, int w, int N, double* dst, int ldd, int real_only, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_cols, dim3(fh_vec_nblk(N, ldd)), dim3(FH_BLOCK), 0, st, src, lds, c0, w, (size_t)N * ldd, dst, ldd, real_only);
}

// ---------------------------------------------------------------------------------------
// dst = [extra +] sum_e w[e] X[e]   (fixed summation order => bitwise reproducible)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(FH_BLOCK) void k_accumulate(const cplx* __restrict__ X, size_t node_stride,
                                                          const cplx* __restrict__ w, int nodes, size_t total,
                                                          cplx* __restrict__ dst, int real_part,
                                                          const cplx* __restrict__ extra) {
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        cplx acc = extra ? extra[e] : cmake(0, 0);
        for (int n = 0; n < nodes; ++n) cfma(acc, w[n], X[(size_t)n * node_stride + e]);
        if (real_part) acc.y = 0.0;
        dst[e] = acc;
    }
}
void fh_launch_accumulate(const cplx* X, size_t node_stride, const cplx* w, int nodes, int N, int ld, const cplx* extra,
                          cplx* dst, int real_part, hipStream_t st) {
    size_t total = (size_t)N * ld;
    int nblk = (int)std::min<size_t>((total + FH_BLOCK - 1) / FH_BLOCK, 2048);
    hipLaunchKernelGGL(k_accumulate, dim3(nblk), dim3(FH_BLOCK), 0, st, X, node_stride, w, nodes, total, dst, real_part, extra);
}

// ---------------------------------------------------------------------------------------
// per-column dots / scaling
// ---------------------------------------------------------------------------------------
int fh_vec_nblk(int N, int ld) {
    size_t total = (size_t)N * ld;
    size_t nb = (total + FH_BLOCK * 4 - 1) / (FH_BLOCK * 4);
    if (nb > 256) nb = 256;
    if (nb < 8) nb = 8;
    return (int)((nb + 7) / 8 * 8);
}

// blocks per node for the batched Krylov kernels: enough blocks to fill 256 CUs, few enough
// partial sums that the per-node finalize kernels stay short (256 per node at >= 8 nodes).
int fh_kry_nblk(int N, int ld, int nodes) {
    size_t total = (size_t)N * ld;
    size_t nb = (total + FH_BLOCK * 4 - 1) / (FH_BLOCK * 4);
    size_t cap = 2048 / (size_t)(nodes < 1 ? 1 : (nodes > 8 ? 8 : nodes));
    if (cap > 512) cap = 512;      // 1-2 local nodes (8-GPU runs): the finalize kernel walks every partial row
    if (nb > cap) nb = cap;
    if (nb < 8) nb = 8;
    return (int)((nb + 7) / 8 * 8);
}

template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_dot_cols(const cplx* __restrict__ U, const cplx* __restrict__ V,
                                                        size_t total, cplx* __restrict__ partial) {
    __shared__ cplx red[FH_BLOCK];
    cplx d = cmake(0, 0);
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK)
        d = cadd(d, cmulc(U[e], V[e]));
    const int t = threadIdx.x;
    red[t] = d;
    __syncthreads();
    if (t < LD) {
        cplx s = red[t];
        for (int k = 1; k < FH_BLOCK / LD; ++k) s = cadd(s, red[t + k * LD]);
        partial[(size_t)blockIdx.x * LD + t] = s;
    }
}
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_sum_partials(const cplx* __restrict__ partial, int nblk,
                                                            cplx* __restrict__ out) {
    __shared__ cplx red[FH_BLOCK];
    const int t = threadIdx.x, c = t % LD, g = t / LD;
    constexpr int G = FH_BLOCK / LD;
    cplx s = cmake(0, 0);
    for (int b = g; b < nblk; b += G) s = cadd(s, partial[(size_t)b * LD + c]);
    red[t] = s;
    __syncthreads();
    if (t < LD) {
        cplx tot = red[t];
        for (int k = 1; k < G; ++k) tot = cadd(tot, red[t + k * LD]);
        out[t] = tot;
    }
}
void fh_launch_dot_cols(const cplx* U, const cplx* V, int N, int ld, cplx* work, cplx* out, hipStream_t st) {
    int nblk = fh_vec_nblk(N, ld);
    size_t total = (size_t)N * ld;
    if (ld == 16) {
        hipLaunchKernelGGL((k_dot_cols<16>), dim3(nblk), dim3(FH_BLOCK), 0, st, U, V, total, work);
        hipLaunchKernelGGL((k_sum_partials<16>), dim3(1), dim3(FH_BLOCK), 0, st, work, nblk, out);
    } else if (ld == 32) {
        hipLaunchKernelGGL((k_dot_cols<32>), dim3(nblk), dim3(FH_BLOCK), 0, st, U, V, total, work);
        hipLaunchKernelGGL((k_sum_partials<32>), dim3(1), dim3(FH_BLOCK), 0, st, work, nblk, out);
    } else {
        hipLaunchKernelGGL((k_dot_cols<64>), dim3(nblk), dim3(FH_BLOCK), 0, st, U, V, total, work);
        hipLaunchKernelGGL((k_sum_partials<64>), dim3(1), dim3(FH_BLOCK), 0, st, work, nblk, out);
    }
}

__global__ __launch_bounds__(FH_BLOCK) void k_scale_cols(cplx* __restrict__ X, const cplx* __restrict__ s,
                                                          size_t total, int ld) {
    const cplx f = s[threadIdx.x % ld];
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK)
        X[e] = cmul(X[e], f);
}
void fh_launch_scale_cols(cplx* X, const cplx* s, int N, int ld, hipStream_t st) {
    hipLaunchKernelGGL(k_scale_cols, dim3(fh_vec_nblk(N, ld)), dim3(FH_BLOCK), 0, st, X, s, (size_t)N * ld, ld);
}

// X[:, c] /= sqrt(dots[c].x) for c < M (columns with a zero norm and columns >= M are left alone): the normalisation of the
// Ritz vectors without a host round trip for the norms
__global__ __launch_bounds__(FH_BLOCK) void k_normalize_cols(cplx* __restrict__ X, const cplx* __restrict__ dots,
                                                              size_t total, int ld, int M) {
    const int c = threadIdx.x % ld;
    const double n2 = dots[c].x;
    if (c >= M || !(n2 > 0.0)) return;
    const double f = 1.0 / sqrt(n2);
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        cplx v = X[e];
        X[e] = cmake(v.x * f, v.y * f);
    }
}
void fh_launch_normalize_cols(cplx* X, const cplx* dots, int N, int ld, int M, hipStream_t st) {
    hipLaunchKernelGGL(k_normalize_cols, dim3(fh_vec_nblk(N, ld)), dim3(FH_BLOCK), 0, st, X, dots, (size_t)N * ld, ld, M);
}

__global__ __launch_bounds__(FH_BLOCK) void k_gather_cols(const cplx* __restrict__ src, const int* __restrict__ perm,
                                                           int count, size_t total, int ld, cplx* __restrict__ dst) {
    const int c = threadIdx.x % ld;
    const int sc = (c < count) ? perm[c] : -1;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        size_t row = e / ld;
        dst[e] = (sc >= 0) ? src[row * ld + sc] : cmake(0, 0);
    }
}
void fh_launch_gather_cols(const cplx* src, const int* perm, int count, int N, int ld, cplx* dst, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_cols, dim3(fh_vec_nblk(N, ld)), dim3(FH_BLOCK), 0, st, src, perm, count,
                       (size_t)N * ld, ld, dst);
}

// ---------------------------------------------------------------------------------------
// Tall-skinny Gram  G = X^H Y  (or X^T Y) on v_mfma_f64_16x16x4_f64.
//   A operand (16 x 4): A[c1][k] = X[i0+k][c1base + c1]   lane l: c1 = l&15, k = l>>4
//   B operand (4 x 16): B[k][c2] = Y[i0+k][c2base + c2]   lane l: c2 = l&15, k = l>>4
//   D (16 x 16): lane l holds D[(l>>4) + 4r][l&15], r = 0..3
// Complex product from four real MFMAs (rr, ii, ri, ir) combined at the end:
//   X^H Y: re = rr + ii, im = ri - ir ;  X^T Y: re = rr - ii, im = ri + ir.
// Each wave owns one 16-column stripe c1 of G and all ld/16 tiles c2 of it; with ld < 64 the
// spare waves take disjoint row subsets.  Per-(block,subset) partial G tiles are summed by a
// second kernel in a fixed order.
// ---------------------------------------------------------------------------------------
#define FH_GRAM_BLOCKS 256

template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_gram_mfma(const cplx* __restrict__ X, const cplx* __restrict__ Y,
                                                         int N, cplx* __restrict__ partial) {
    constexpr int NS = LD / 16;        // stripes (= tiles per stripe)
    constexpr int SUB = 4 / NS;        // row subsets per block (waves per stripe)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stripe = wave % NS, sub = wave / NS;
    const int lc = lane & 15, lk = lane >> 4;
    const int rows_per_block = ((N + gridDim.x - 1) / gridDim.x + 3) & ~3;
    const int row_begin = blockIdx.x * rows_per_block;
    const int row_end = min(N, row_begin + rows_per_block);

    v4d rr[NS], ii[NS], ri[NS], ir[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) { rr[t] = (v4d){0, 0, 0, 0}; ii[t] = rr[t]; ri[t] = rr[t]; ir[t] = rr[t]; }

    // four row groups per step with all their loads issued first (the loop was one load -> 16 MFMAs -> next load, i.e.
    // bound by the load latency); the accumulation order per wave is unchanged, so results are bit-identical
    constexpr int UG = 4;
    for (int i0 = row_begin + sub * 4; i0 < row_end; i0 += 4 * SUB * UG) {
        cplx x[UG], y[UG][NS];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int i = i0 + u * 4 * SUB + lk;
            const bool in = i < row_end;
            x[u] = in ? X[(size_t)i * LD + stripe * 16 + lc] : cmake(0, 0);
#pragma unroll
            for (int t = 0; t < NS; ++t) y[u][t] = in ? Y[(size_t)i * LD + t * 16 + lc] : cmake(0, 0);
        }
        // Panels without imaginary parts (the real projection of a real pencil: Q_proj, its orthonormal basis, A Q, the
        // Ritz vectors) need one of the four real products; the skipped ones would add exact zeros, so the result is
        // bit-identical.  Decided per step from the data the wave just loaded, not from a mode flag.
        bool has_imag = false;
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            has_imag |= x[u].y != 0.0;
#pragma unroll
            for (int t = 0; t < NS; ++t) has_imag |= y[u][t].y != 0.0;
        }
        if (__any(has_imag)) {
#pragma unroll
            for (int u = 0; u < UG; ++u) {
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    rr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].x, y[u][t].x, rr[t], 0, 0, 0);
                    ii[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].y, y[u][t].y, ii[t], 0, 0, 0);
                    ri[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].x, y[u][t].y, ri[t], 0, 0, 0);
                    ir[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].y, y[u][t].x, ir[t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < UG; ++u) {
#pragma unroll
                for (int t = 0; t < NS; ++t) rr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].x, y[u][t].x, rr[t], 0, 0, 0);
            }
        }
    }
    // partial layout: [block][sub][c2 (col-major G: c1 + LD*c2)] raw four-accumulator form
    // is combined here with sign = +1 (conj) stored as (rr, ii, ri, ir) -> two cplx per entry
    cplx* p = partial + ((size_t)blockIdx.x * SUB + sub) * (size_t)LD * LD * 2;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int c1 = stripe * 16 + lk + 4 * r;
            int c2 = t * 16 + lc;
            size_t o = ((size_t)c2 * LD + c1) * 2;
            p[o] = cmake(rr[t][r], ii[t][r]);
            p[o + 1] = cmake(ri[t][r], ir[t][r]);
        }
    }
}

// Sum of the per-(block, subset) partial tiles in a fixed order, two stages: FH_GRAM_GROUPS groups of consecutive slots
// are summed side by side (grid.y), then the group sums in group order.  (One stage with a thread per entry walked
// 256 slots of 131 KB stride from 16 workgroups: about half of the whole Gram product's time.)
#define FH_GRAM_GROUPS 16
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_gram_reduce_groups(const cplx* __restrict__ partial, int nslots,
                                                                  cplx* __restrict__ gsum) {
    const int e = blockIdx.x * FH_BLOCK + threadIdx.x;
    if (e >= LD * LD) return;
    const int per = (nslots + FH_GRAM_GROUPS - 1) / FH_GRAM_GROUPS;
    const int s0 = blockIdx.y * per, s1 = min(nslots, s0 + per);
    double rr = 0, ii = 0, ri = 0, ir = 0;
    int s = s0;
    for (; s + 8 <= s1; s += 8) {            // eight slots' loads in flight, added in slot order
        cplx a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const cplx* p = partial + (size_t)(s + q) * LD * LD * 2 + (size_t)e * 2;
            a[q] = p[0]; b[q] = p[1];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { rr += a[q].x; ii += a[q].y; ri += b[q].x; ir += b[q].y; }
    }
    for (; s < s1; ++s) {
        const cplx* p = partial + (size_t)s * LD * LD * 2 + (size_t)e * 2;
        cplx a = p[0], b = p[1];
        rr += a.x; ii += a.y; ri += b.x; ir += b.y;
    }
    cplx* o = gsum + ((size_t)blockIdx.y * LD * LD + e) * 2;
    o[0] = cmake(rr, ii);
    o[1] = cmake(ri, ir);
}
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_gram_reduce(const cplx* __restrict__ gsum, int bilinear, cplx* __restrict__ G) {
    const int e = blockIdx.x * FH_BLOCK + threadIdx.x;
    if (e >= LD * LD) return;
    cplx a[FH_GRAM_GROUPS], b[FH_GRAM_GROUPS];
#pragma unroll
    for (int g = 0; g < FH_GRAM_GROUPS; ++g) {
        const cplx* p = gsum + ((size_t)g * LD * LD + e) * 2;
        a[g] = p[0]; b[g] = p[1];
    }
    double rr = 0, ii = 0, ri = 0, ir = 0;
#pragma unroll
    for (int g = 0; g < FH_GRAM_GROUPS; ++g) { rr += a[g].x; ii += a[g].y; ri += b[g].x; ir += b[g].y; }
    G[e] = bilinear ? cmake(rr - ii, ri + ir) : cmake(rr + ii, ri - ir);
}

size_t fh_gram_work_elems(int ld) {
    int sub = 4 / (ld / 16);
    return (size_t)FH_GRAM_BLOCKS * sub * ld * ld * 2 + (size_t)FH_GRAM_GROUPS * ld * ld * 2;
}

template <int LD>
static void gram_launch(const cplx* X, const cplx* Y, int N, int bilinear, cplx* work, cplx* G, hipStream_t st) {
    constexpr int sub = 4 / (LD / 16), nslots = FH_GRAM_BLOCKS * sub;
    const int nred = (LD * LD + FH_BLOCK - 1) / FH_BLOCK;
    cplx* gsum = work + (size_t)nslots * LD * LD * 2;
    hipLaunchKernelGGL((k_gram_mfma<LD>), dim3(FH_GRAM_BLOCKS), dim3(FH_BLOCK), 0, st, X, Y, N, work);
    hipLaunchKernelGGL((k_gram_reduce_groups<LD>), dim3(nred, FH_GRAM_GROUPS), dim3(FH_BLOCK), 0, st, work, nslots, gsum);
    hipLaunchKernelGGL((k_gram_reduce<LD>), dim3(nred), dim3(FH_BLOCK), 0, st, gsum, bilinear, G);
}
void fh_launch_gram(const cplx* X, const cplx* Y, int N, int ld, int bilinear, cplx* work, cplx* G,
                    hipStream_t st) {
    if (ld == 16) gram_launch<16>(X, Y, N, bilinear, work, G, st);
    else if (ld == 32) gram_launch<32>(X, Y, N, bilinear, work, G, st);
    else gram_launch<64>(X, Y, N, bilinear, work, G, st);
}

// ---------------------------------------------------------------------------------------
// Xout = Q * V  (N x ld) * (ld x ld): V staged in LDS, one output element per thread.
// ---------------------------------------------------------------------------------------
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_small_matmul(const cplx* __restrict__ Q, const cplx* __restrict__ V,
                                                            int N, cplx* __restrict__ Xout) {
    extern __shared__ cplx sm[];
    cplx* Vs = sm;                 // [k][c]  (row k of V contiguous over c) : LD*LD
    cplx* Qs = sm + LD * LD;       // [rows][LD] tile of Q : (256/LD) * LD = 256
    const int t = threadIdx.x;
    for (int e = t; e < LD * LD; e += FH_BLOCK) {
        int k = e / LD, c = e % LD;
        Vs[e] = V[(size_t)c * LD + k];          // V column-major: V[k + LD*c]
    }
    constexpr int RPB = FH_BLOCK / LD;
    const int c = t % LD, r = t / LD;
    __syncthreads();
    for (int i0 = blockIdx.x * RPB; i0 < N; i0 += gridDim.x * RPB) {
        const int i = i0 + r;
        Qs[t] = (i < N) ? Q[(size_t)i * LD + c] : cmake(0, 0);
        __syncthreads();
        cplx acc = cmake(0, 0);
#pragma unroll 8
        for (int k = 0; k < LD; ++k) cfma(acc, Qs[r * LD + k], Vs[k * LD + c]);
        if (i < N) Xout[(size_t)i * LD + c] = acc;
        __syncthreads();
    }
}
// MFMA form (default): a wave owns one 16-column tile of the output and keeps its slice of V (ld x 16) in registers
// as the B operand for the whole kernel; rows of Q go through LDS 16 at a time (coalesced load, read back as the A
// operand), the product is written row-major in 256 B segments.  The VALU form above reads two LDS operands per
// FMA and is LDS-bound (85 us for 50 000 x 64 against 25 us of HBM time).
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_small_matmul_mfma(const cplx* __restrict__ Q, const cplx* __restrict__ V,
                                                                 int N, cplx* __restrict__ Xout) {
    constexpr int NS = LD / 16, SUB = 4 / NS, RB = 16 * SUB, KS = LD / 4;
    __shared__ cplx Qs[RB][LD + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int ct = wave % NS, sb = wave / NS;
    cplx vB[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) vB[kk] = V[(size_t)(16 * ct + lr) * LD + 4 * kk + lk];     // V[k + LD*c]
    // real V (real reduced eigenvectors, real R^-1) and real rows of Q: one real product instead of four, bit-identical
    bool v_imag = false;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) v_imag |= vB[kk].y != 0.0;
    v_imag = __any(v_imag);
    for (int i0 = blockIdx.x * RB; i0 < N; i0 += gridDim.x * RB) {
        int q_imag = 0;
        for (int e = t; e < RB * LD; e += FH_BLOCK) {
            const int r = e / LD, c = e % LD;
            const cplx q = (i0 + r < N) ? Q[(size_t)(i0 + r) * LD + c] : cmake(0, 0);
            q_imag |= q.y != 0.0;
            Qs[r][c] = q;
        }
        q_imag = __syncthreads_or(q_imag);          // the barrier the tile needs anyway
        v4d re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
        if (v_imag || q_imag) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const cplx a = Qs[16 * sb + lr][4 * kk + lk];
                re = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, vB[kk].x, re, 0, 0, 0);
                re = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, vB[kk].y, re, 0, 0, 0);
                im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, vB[kk].y, im, 0, 0, 0);
                im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, vB[kk].x, im, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) re = __builtin_amdgcn_mfma_f64_16x16x4f64(Qs[16 * sb + lr][4 * kk + lk].x, vB[kk].x, re, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 16 * sb + lk + 4 * r;
            if (i < N) Xout[(size_t)i * LD + 16 * ct + lr] = cmake(re[r], im[r]);
        }
        __syncthreads();
    }
}

void fh_launch_small_matmul(const cplx* Q, const cplx* V, int N, int ld, cplx* Xout, hipStream_t st) {
    static const bool valu = getenv("FH_SMALL_MATMUL_VALU") != nullptr;
    if (!valu) {
        const int rb = 16 * (4 / (ld / 16));
        const int nb = std::min((N + rb - 1) / rb, 2048);
        if (ld == 16) hipLaunchKernelGGL((k_small_matmul_mfma<16>), dim3(nb), dim3(FH_BLOCK), 0, st, Q, V, N, Xout);
        else if (ld == 32) hipLaunchKernelGGL((k_small_matmul_mfma<32>), dim3(nb), dim3(FH_BLOCK), 0, st, Q, V, N, Xout);
        else hipLaunchKernelGGL((k_small_matmul_mfma<64>), dim3(nb), dim3(FH_BLOCK), 0, st, Q, V, N, Xout);
        return;
    }
    int rpb = FH_BLOCK / ld;
    int nblk = std::min((N + rpb - 1) / rpb, 2048);
    size_t shm = ((size_t)ld * ld + FH_BLOCK) * sizeof(cplx);
    if (ld == 16) hipLaunchKernelGGL((k_small_matmul<16>), dim3(nblk), dim3(FH_BLOCK), shm, st, Q, V, N, Xout);
    else if (ld == 32) hipLaunchKernelGGL((k_small_matmul<32>), dim3(nblk), dim3(FH_BLOCK), shm, st, Q, V, N, Xout);
    else hipLaunchKernelGGL((k_small_matmul<64>), dim3(nblk), dim3(FH_BLOCK), shm, st, Q, V, N, Xout);
}

// ---------------------------------------------------------------------------------------
// Column-pivoted modified Gram-Schmidt with one re-orthogonalisation pass, device-driven.
// Same pivot rule and rank rule as LAPACK ZGEQP3 + _feast_qr_compress!:
//   pivot = remaining column of largest norm; R_kk = that norm;
//   rank  = #{ k : R_kk > max(rank_tol, eps*max(N,m)) * R_11 }, stopping at the first failure.
// istate = {k, rank, done, pivot, perm[ld]},  dstate = {R11, thr, norm2[ld]}.
// Columns are never moved during the factorisation; `perm` records the pivot order and the
// caller gathers the first `rank` pivots into the output basis.
// ---------------------------------------------------------------------------------------
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_mgs_norms(const cplx* __restrict__ X, size_t total,
                                                         cplx* __restrict__ partial) {
    __shared__ cplx red[FH_BLOCK];
    cplx d = cmake(0, 0);
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK)
        d.x += cabs2(X[e]);
    const int t = threadIdx.x;
    red[t] = d;
    __syncthreads();
    if (t < LD) {
        cplx s = red[t];
        for (int k = 1; k < FH_BLOCK / LD; ++k) s = cadd(s, red[t + k * LD]);
        partial[(size_t)blockIdx.x * LD + t] = s;
    }
}

// choose next pivot from column norms (sum of partials); one block
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_mgs_pick(const cplx* __restrict__ partial, int nblk, int N, int m,
                                                        double rank_tol, double ref_scale, int big_dim, int* istate,
                                                        double* dstate) {
    __shared__ cplx red[FH_BLOCK];
    __shared__ double nrm[LD];
    __shared__ int used[LD];
    const int t = threadIdx.x, c = t % LD, g = t / LD;
    constexpr int G = FH_BLOCK / LD;
    if (istate[2]) return;   // done
    cplx s = cmake(0, 0);
    for (int b = g; b < nblk; b += G) s = cadd(s, partial[(size_t)b * LD + c]);
    red[t] = s;
    if (t < LD) used[t] = 0;
    __syncthreads();
    // columns already chosen: perm[0..k) -- marked by k threads (one thread walking the list per candidate column was
    // 4 096 dependent global loads at k = 64: 50 of this kernel's 66 us)
    if (t < istate[0] && t < LD) used[istate[4 + t]] = 1;
    if (t < LD) {
        double tot = red[t].x;
        for (int k = 1; k < G; ++k) tot += red[t + k * LD].x;
        nrm[t] = tot;
        dstate[2 + t] = tot;
    }
    __syncthreads();
    if (t == 0) {
        const int k = istate[0];
        double best = -1.0; int bp = -1;
        for (int cc = 0; cc < m; ++cc) {
            if (used[cc]) continue;
            if (nrm[cc] > best) { best = nrm[cc]; bp = cc; }
        }
        double rkk = (bp >= 0) ? sqrt(best) : 0.0;
        if (k == 0) {
            // ref_scale > 0: this panel is a block of a wider matrix whose largest column norm
            // (the R_11 of the reference's pivoted QR) was found by the caller
            const double r11 = rkk > ref_scale ? rkk : ref_scale;
            dstate[0] = r11;
            double eps = 2.220446049250313e-16;
            double big = (double)(N > big_dim ? N : big_dim);
            double th = rank_tol > eps * big ? rank_tol : eps * big;
            dstate[1] = th * r11;
        }
        if (bp < 0 || !(rkk > dstate[1]) || rkk == 0.0) {
            istate[2] = 1;           // done: rank = k
            istate[1] = k;
        } else {
            istate[3] = bp;
            istate[4 + k] = bp;
        }
    }
}

// d[c] = <x_p, x_c> partials
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_mgs_dot(const cplx* __restrict__ X, int N, const int* istate,
                                                       cplx* __restrict__ partial) {
    __shared__ cplx red[FH_BLOCK];
    const int t = threadIdx.x, c = t % LD, r = t / LD;
    constexpr int RPB = FH_BLOCK / LD;
    cplx d = cmake(0, 0);
    if (!istate[2]) {
        const int p = istate[3];
        for (int i = blockIdx.x * RPB + r; i < N; i += gridDim.x * RPB) {
            cplx xp = X[(size_t)i * LD + p];
            cplx xc = X[(size_t)i * LD + c];
            d = cadd(d, cmulc(xp, xc));
        }
    }
    red[t] = d;
    __syncthreads();
    if (t < LD) {
        cplx s = red[t];
        for (int k = 1; k < RPB; ++k) s = cadd(s, red[t + k * LD]);
        partial[(size_t)blockIdx.x * LD + t] = s;
    }
}

// coef[c] = d_c / d_p for columns not yet chosen (else 0); coef[p] encodes the scale:
// on the final pass of a step the pivot is normalised by 1/sqrt(d_p).
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_mgs_coef(const cplx* __restrict__ partial, int nblk, int m,
                                                        const int* istate, int normalize, cplx* coef) {
    __shared__ cplx red[FH_BLOCK];
    __shared__ cplx dsum[LD];
    __shared__ int taken[LD];
    const int t = threadIdx.x, c = t % LD, g = t / LD;
    constexpr int G = FH_BLOCK / LD;
    if (istate[2]) return;
    cplx s = cmake(0, 0);
    for (int b = g; b < nblk; b += G) s = cadd(s, partial[(size_t)b * LD + c]);
    red[t] = s;
    if (t < LD) taken[t] = 0;
    __syncthreads();
    if (t <= istate[0] && t < LD) taken[istate[4 + t]] = 1;                  // chosen columns, the pivot itself included
    if (t < LD) {
        cplx tot = red[t];
        for (int k = 1; k < G; ++k) tot = cadd(tot, red[t + k * LD]);
        dsum[t] = tot;
    }
    __syncthreads();
    if (t < LD) {
        const int p = istate[3];
        const double dp = dsum[p].x;
        const bool used = (t >= m) || taken[t] != 0;
        cplx cf = cmake(0, 0);
        if (!used && dp > 0.0) cf = cmake(dsum[t].x / dp, dsum[t].y / dp);
        if (t == p) cf = cmake(normalize && dp > 0.0 ? 1.0 / sqrt(dp) : 1.0, 0.0);
        coef[t] = cf;
    }
}

// x_c -= coef[c] x_p (c != p);  x_p *= coef[p];  partial norms^2 of the updated columns
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_mgs_update(cplx* __restrict__ X, int N, const int* istate,
                                                          const cplx* __restrict__ coef, cplx* __restrict__ partial) {
    __shared__ cplx red[FH_BLOCK];
    const int t = threadIdx.x, c = t % LD, r = t / LD;
    constexpr int RPB = FH_BLOCK / LD;
    cplx d = cmake(0, 0);
    if (!istate[2]) {
        const int p = istate[3];
        const cplx cf = coef[c];
        for (int i = blockIdx.x * RPB + r; i < N; i += gridDim.x * RPB) {
            cplx xp = X[(size_t)i * LD + p];
            cplx xc = X[(size_t)i * LD + c];
            // all lanes of the row read x_p before any lane overwrites it: the row is handled
            // by LD consecutive lanes of one wave (LD <= 64), executing in lock-step.
            cplx nv = (c == p) ? cmul(xc, cf) : csub(xc, cmul(cf, xp));
            X[(size_t)i * LD + c] = nv;
            d.x += cabs2(nv);
        }
    }
    red[t] = d;
    __syncthreads();
    if (t < LD) {
        cplx s = red[t];
        for (int k = 1; k < RPB; ++k) s = cadd(s, red[t + k * LD]);
        partial[(size_t)blockIdx.x * LD + t] = s;
    }
}

__global__ void k_mgs_advance(int* istate, int m) {
    if (threadIdx.x == 0 && !istate[2]) {
        istate[0] += 1;
        if (istate[0] >= m) { istate[2] = 1; istate[1] = m; }
    }
}
__global__ void k_mgs_init(int* istate, int ld) {
    if (threadIdx.x < 4 + ld) istate[threadIdx.x] = (threadIdx.x >= 4) ? -1 : 0;
}

template <int LD>
static void mgs_run_ld(const fh_mgs_args& a, hipStream_t st) {
    const int RPB = FH_BLOCK / LD;
    int nblk = std::min((a.N + RPB * 4 - 1) / (RPB * 4), 256);
    if (nblk < 1) nblk = 1;
    int nblk_flat = fh_vec_nblk(a.N, LD);
    size_t total = (size_t)a.N * LD;
    hipLaunchKernelGGL(k_mgs_init, dim3(1), dim3(128), 0, st, a.istate, LD);
    hipLaunchKernelGGL((k_mgs_norms<LD>), dim3(nblk_flat), dim3(FH_BLOCK), 0, st, a.X, total, a.work);
    hipLaunchKernelGGL((k_mgs_pick<LD>), dim3(1), dim3(FH_BLOCK), 0, st, a.work, nblk_flat, a.N, a.m, a.rank_tol,
                       a.ref_scale, a.big_dim > a.m ? a.big_dim : a.m, a.istate, a.dstate);
    for (int k = 0; k < a.m; ++k) {
        // first projection pass (no normalisation), then re-orthogonalise and normalise pivot
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL((k_mgs_dot<LD>), dim3(nblk), dim3(FH_BLOCK), 0, st, a.X, a.N, a.istate, a.work);
            hipLaunchKernelGGL((k_mgs_coef<LD>), dim3(1), dim3(FH_BLOCK), 0, st, a.work, nblk, a.m, a.istate,
                               pass == 1 ? 1 : 0, a.coef);
            hipLaunchKernelGGL((k_mgs_update<LD>), dim3(nblk), dim3(FH_BLOCK), 0, st, a.X, a.N, a.istate, a.coef,
                               a.work);
        }
        hipLaunchKernelGGL(k_mgs_advance, dim3(1), dim3(64), 0, st, a.istate, a.m);
        // norms of the remaining columns come from the last update's partials
        hipLaunchKernelGGL((k_mgs_pick<LD>), dim3(1), dim3(FH_BLOCK), 0, st, a.work, nblk, a.N, a.m, a.rank_tol,
                           a.ref_scale, a.big_dim > a.m ? a.big_dim : a.m, a.istate, a.dstate);
    }
}

void fh_mgs_run(const fh_mgs_args& a, hipStream_t st) {
    if (a.ld == 16) mgs_run_ld<16>(a, st);
    else if (a.ld == 32) mgs_run_ld<32>(a, st);
    else mgs_run_ld<64>(a, st);
}
