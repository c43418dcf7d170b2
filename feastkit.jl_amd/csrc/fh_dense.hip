// fh_dense.hip -- dense shifted systems for the FEAST contour sweep (gfx950).
//
// Replaces  @. shifted = z*B - A        src/dense/feast_dense.jl:193, src/core/feast_aux.jl:59-74
//           lu(shifted_matrix)          src/dense/feast_dense.jl:196   (ZGETRF, partial pivoting)
//           ldiv!(solutions, factor, rhs) src/dense/feast_dense.jl:207 (ZGETRS, M0 right-hand sides)
//           mul!(rhs, B, basis) / mul!(aq_work, A, q_rank)  :184, :252  (tall-skinny products)
// for all local quadrature nodes in one batch (node = grid.y / grid.x of each launch).
//
// Factors are N x N c128 column-major (LAPACK layout, so pivot search walks contiguous memory);
// right-hand sides / solutions are row-major N x ld panels (fh_common.hpp), so a row
// interchange moves one contiguous line and the triangular sweeps stream whole rows.
#include <stdlib.h>
#include <chrono>
#include "fh_common.hpp"
#include "fh_kernels.hpp"
#include "fh_dense.hpp"
#include "../../include/feasthip.h"

#define FH_BLOCK 256
#define LU_NB 32
#define LU_PANEL_THREADS 1024

// ---------------------------------------------------------------------------------------
// dense operator on panels:  Y = (Bvec -) (cb*B + ca*A) X  with the same fused dots as k_spmm
// ---------------------------------------------------------------------------------------
int fh_dense_op_nblk(int N) { return (N + 63) / 64; }

template <typename VT, int LD, bool BIDENT>
__global__ __launch_bounds__(FH_BLOCK) void k_dense_op(fh_dense_op_args a) {
    constexpr int TR = 64, TJ = 16, CPT = LD / 4;
    __shared__ VT As[TJ][TR];
    __shared__ VT Bs[BIDENT ? 1 : TJ][BIDENT ? 1 : TR];
    __shared__ cplx Xs[TJ][LD];
    __shared__ cplx red[FH_BLOCK];
    const int node = blockIdx.y;
    const int t = threadIdx.x;
    const bool skip = a.node_active && a.node_active[node] == 0;
    const int N = a.N;
    const int i0 = blockIdx.x * TR;
    const int r = t % TR, cg = t / TR;
    const cplx* X = a.X + (size_t)node * a.x_node_stride;
    cplx* Y = a.Y + (size_t)node * a.y_node_stride;
    const VT* A = (const VT*)a.A;
    const VT* B = (const VT*)a.B;
    cplx accA[CPT], accB[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) { accA[k] = cmake(0, 0); accB[k] = cmake(0, 0); }
    if (!skip) {
        for (int j0 = 0; j0 < N; j0 += TJ) {
#pragma unroll
            for (int q = 0; q < TJ / 4; ++q) {
                int jj = t / TR + 4 * q;
                int i = i0 + r, j = j0 + jj;
                bool ok = (i < N && j < N);
                if constexpr (sizeof(VT) == sizeof(cplx)) {
                    As[jj][r] = ok ? A[(size_t)j * N + i] : VT{0, 0};
                    if (!BIDENT) Bs[jj][r] = ok ? B[(size_t)j * N + i] : VT{0, 0};
                } else {
                    As[jj][r] = ok ? A[(size_t)j * N + i] : VT(0);
                    if (!BIDENT) Bs[jj][r] = ok ? B[(size_t)j * N + i] : VT(0);
                }
            }
            for (int e = t; e < TJ * LD; e += FH_BLOCK) {
                int jj = e / LD, c = e % LD;
                Xs[jj][c] = (j0 + jj < N) ? X[(size_t)(j0 + jj) * LD + c] : cmake(0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < TJ; ++jj) {
                VT av = As[jj][r];
                VT bv = av;
                if (!BIDENT) bv = Bs[jj][r];
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    cplx x = Xs[jj][cg * CPT + k];
                    cplx pa = vmul(av, x);
                    accA[k] = cadd(accA[k], pa);
                    if (!BIDENT) { cplx pb = vmul(bv, x); accB[k] = cadd(accB[k], pb); }
                }
            }
            __syncthreads();
        }
    }
    const int i = i0 + r;
    cplx d1[CPT], d2[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        d1[k] = cmake(0, 0); d2[k] = cmake(0, 0);
        const int c = cg * CPT + k;
        if (!skip && i < N) {
            const cplx ca = a.coefA[node * LD + c], cb = a.coefB[node * LD + c];
            cplx y = cmul(ca, accA[k]);
            cplx xown = cmake(0, 0);
            if (BIDENT || a.dot_mode == 2 || a.dot_mode == 4) xown = X[(size_t)i * LD + c];
            if (BIDENT) cfma(y, cb, xown); else cfma(y, cb, accB[k]);
            if (a.Bvec) y = csub(a.Bvec[(size_t)node * a.b_node_stride + (size_t)i * LD + c], y);
            Y[(size_t)i * LD + c] = y;
            if (a.dot_mode == 1) d1[k] = cmulc(a.U[(size_t)node * a.u_node_stride + (size_t)i * LD + c], y);
            else if (a.dot_mode == 2) { d1[k] = cmulc(y, xown); d2[k].x = cabs2(y); }
            else if (a.dot_mode == 3) d2[k].x = cabs2(y);
            else if (a.dot_mode == 4) d1[k] = cmul(xown, y);
        }
    }
    if (a.dot_mode != 0) {
        // reduce over the 64 rows of the block for each column: thread (r, cg) holds CPT columns
        const size_t o = ((size_t)node * gridDim.x + blockIdx.x) * LD;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 0 && !(a.dot_mode == 1 || a.dot_mode == 2 || a.dot_mode == 4)) continue;
            if (pass == 1 && !(a.dot_mode == 2 || a.dot_mode == 3)) continue;
            for (int k = 0; k < CPT; ++k) {
                red[t] = pass == 0 ? d1[k] : d2[k];
                __syncthreads();
                if (r == 0) {
                    cplx s = cmake(0, 0);
                    for (int q = 0; q < TR; ++q) s = cadd(s, red[cg * TR + q]);
                    (pass == 0 ? a.partial1 : a.partial2)[o + cg * CPT + k] = s;
                }
                __syncthreads();
            }
        }
    }
}

// Plain products (dot_mode 0: project, Ritz residual, refinement residual) run on the f64 matrix
// cores: one wave owns a 16-row band of the output and TPW 16-column tiles, the contraction is
// staged in KC-deep chunks with the panel rows shared through LDS (split re/im, rows padded by
// 16 doubles so the four k-rows of an operand read fall into different bank halves) and the matrix
// operands taken straight from global memory (16 consecutive rows of one column = one 128/256 B
// segment per 16 lanes).  Blocks are numbered so that the 8 consecutive ids that land on the 8 XCDs
// carry 8 different row tiles and the blocks of one XCD walk the nodes of the same row tile, which
// keeps the tile of A in that XCD's L2 while every node consumes it.
typedef double dop_v4d __attribute__((ext_vector_type(4)));
#define DOP_KC 16
#define DOP_PAD 16

template <typename VT, int LD, bool BIDENT, int TRB>
__global__ __launch_bounds__(FH_BLOCK) void k_dense_op_mfma(fh_dense_op_args a, int row_tiles) {
    constexpr int KC = DOP_KC, CT = LD / 16, NCG = 4 / TRB, TPW = CT / NCG, XPT = KC * LD / FH_BLOCK;
    constexpr bool CPLX = sizeof(VT) == sizeof(cplx);
    static_assert(TPW >= 1 && XPT >= 1, "tile split");
    __shared__ double Xre[KC][LD + DOP_PAD];
    __shared__ double Xim[KC][LD + DOP_PAD];
    const int b = blockIdx.x;
    const int rt = (b & 7) + 8 * ((b >> 3) / a.nodes);
    const int node = (b >> 3) % a.nodes;
    if (rt >= row_tiles) return;
    if (a.node_active && a.node_active[node] == 0) return;
    const int N = a.N;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int band = w % TRB, cg = w / TRB;
    const int i0 = rt * (16 * TRB);
    const int irow = i0 + 16 * band + lr;          // matrix row this lane feeds as MFMA A operand
    const cplx* X = a.X + (size_t)node * a.x_node_stride;
    cplx* Y = a.Y + (size_t)node * a.y_node_stride;
    const VT* A = (const VT*)a.A;
    const VT* B = (const VT*)a.B;
    dop_v4d aR[TPW], aI[TPW], bR[BIDENT ? 1 : TPW], bI[BIDENT ? 1 : TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        aR[q] = dop_v4d{0, 0, 0, 0}; aI[q] = dop_v4d{0, 0, 0, 0};
        if (!BIDENT) { bR[q] = dop_v4d{0, 0, 0, 0}; bI[q] = dop_v4d{0, 0, 0, 0}; }
    }
    cplx xn[XPT];
    double are[KC / 4], aim[KC / 4], bre[KC / 4], bim[KC / 4];
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int e = t + q * FH_BLOCK, jj = e / LD, c = e % LD;
            xn[q] = (j0 + jj < N) ? X[(size_t)(j0 + jj) * LD + c] : cmake(0, 0);
        }
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            const int j = j0 + 4 * s + lk;
            const bool ok = irow < N && j < N;
            if constexpr (CPLX) {
                cplx v = ok ? A[(size_t)j * N + irow] : cmake(0, 0);
                are[s] = v.x; aim[s] = v.y;
                if (!BIDENT) { cplx u = ok ? B[(size_t)j * N + irow] : cmake(0, 0); bre[s] = u.x; bim[s] = u.y; }
            } else {
                are[s] = ok ? A[(size_t)j * N + irow] : 0.0;
                if (!BIDENT) bre[s] = ok ? B[(size_t)j * N + irow] : 0.0;
            }
        }
    };
    load_chunk(0);
    for (int j0 = 0; j0 < N; j0 += KC) {
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int e = t + q * FH_BLOCK, jj = e / LD, c = e % LD;
            Xre[jj][c] = xn[q].x; Xim[jj][c] = xn[q].y;
        }
        double cre[KC / 4], cim[KC / 4], dre[KC / 4], dim_[KC / 4];
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) { cre[s] = are[s]; cim[s] = aim[s]; dre[s] = bre[s]; dim_[s] = bim[s]; }
        __syncthreads();
        if (j0 + KC < N) load_chunk(j0 + KC);
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            double xr[TPW], xi[TPW];
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                xr[q] = Xre[4 * s + lk][16 * (cg * TPW + q) + lr];
                xi[q] = Xim[4 * s + lk][16 * (cg * TPW + q) + lr];
            }
#pragma unroll
            for (int q = 0; q < TPW; ++q) aR[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cre[s], xr[q], aR[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TPW; ++q) aI[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cre[s], xi[q], aI[q], 0, 0, 0);
            if constexpr (CPLX) {
#pragma unroll
                for (int q = 0; q < TPW; ++q) aR[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-cim[s], xi[q], aR[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < TPW; ++q) aI[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cim[s], xr[q], aI[q], 0, 0, 0);
            }
            if constexpr (!BIDENT) {
#pragma unroll
                for (int q = 0; q < TPW; ++q) bR[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(dre[s], xr[q], bR[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < TPW; ++q) bI[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(dre[s], xi[q], bI[q], 0, 0, 0);
                if constexpr (CPLX) {
#pragma unroll
                    for (int q = 0; q < TPW; ++q) bR[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-dim_[s], xi[q], bR[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < TPW; ++q) bI[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(dim_[s], xr[q], bI[q], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // D layout of v_mfma_f64_16x16x4: register r of lane (lk, lr) is row lk + 4 r, column lr
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int c = 16 * (cg * TPW + q) + lr;
        const cplx ca = a.coefA[node * LD + c], cb = a.coefB[node * LD + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 16 * band + lk + 4 * r;
            if (i >= N) continue;
            cplx y = cmul(ca, cmake(aR[q][r], aI[q][r]));
            if (BIDENT) cfma(y, cb, X[(size_t)i * LD + c]); else cfma(y, cb, cmake(bR[q][r], bI[q][r]));
            if (a.Bvec) y = csub(a.Bvec[(size_t)node * a.b_node_stride + (size_t)i * LD + c], y);
            Y[(size_t)i * LD + c] = y;
        }
    }
}

template <typename VT, int LD, int TRB>
static void launch_dense_op_mfma(const fh_dense_op_args& a, hipStream_t st) {
    const int row_tiles = (a.N + 16 * TRB - 1) / (16 * TRB);
    const int groups = (row_tiles + 7) / 8;
    dim3 grid(8 * a.nodes * groups), block(FH_BLOCK);
    if (a.B == nullptr) hipLaunchKernelGGL((k_dense_op_mfma<VT, LD, true, TRB>), grid, block, 0, st, a, row_tiles);
    else hipLaunchKernelGGL((k_dense_op_mfma<VT, LD, false, TRB>), grid, block, 0, st, a, row_tiles);
}

template <typename VT, int LD>
static void launch_dense_op_ld(const fh_dense_op_args& a, int nblk, hipStream_t st) {
    static const bool no_mfma = getenv("FH_DENSE_OP_VALU") != nullptr;
    if (a.dot_mode == 0 && !no_mfma) {
        // 32-row tiles when 64-row tiles would leave CUs idle (single-node calls on mid-size matrices)
        if constexpr (LD >= 32) {
            if ((long)a.nodes * ((a.N + 63) / 64) < 256) { launch_dense_op_mfma<VT, LD, 2>(a, st); return; }
        }
        launch_dense_op_mfma<VT, LD, 4>(a, st);
        return;
    }
    dim3 grid(nblk, a.nodes), block(FH_BLOCK);
    if (a.B == nullptr) hipLaunchKernelGGL((k_dense_op<VT, LD, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_dense_op<VT, LD, false>), grid, block, 0, st, a);
}
void fh_launch_dense_op(const fh_dense_op_args& a, int ld, int nblk, hipStream_t st) {
    if (a.is_complex) {
        if (ld == 16) launch_dense_op_ld<cplx, 16>(a, nblk, st);
        else if (ld == 32) launch_dense_op_ld<cplx, 32>(a, nblk, st);
        else launch_dense_op_ld<cplx, 64>(a, nblk, st);
    } else {
        if (ld == 16) launch_dense_op_ld<double, 16>(a, nblk, st);
        else if (ld == 32) launch_dense_op_ld<double, 32>(a, nblk, st);
        else launch_dense_op_ld<double, 64>(a, nblk, st);
    }
}

__global__ __launch_bounds__(FH_BLOCK) void k_axpy_cols(cplx* __restrict__ R, const cplx* __restrict__ X,
                                                         const cplx* __restrict__ lam, size_t total, int ld) {
    const cplx l = lam[threadIdx.x % ld];
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK)
        R[e] = csub(R[e], cmul(l, X[e]));
}
void fh_launch_axpy_cols(cplx* R, const cplx* X, const cplx* lam, int N, int ld, hipStream_t st) {
    hipLaunchKernelGGL(k_axpy_cols, dim3(fh_vec_nblk(N, ld)), dim3(FH_BLOCK), 0, st, R, X, lam, (size_t)N * ld, ld);
}

// ---------------------------------------------------------------------------------------
// batched LU -- templated on the factor element type T: cplx (complex128) or cplxf (complex64,
// factor_precision = 32: factors and triangular solves in single precision inside an fp64
// iterative-refinement loop, fh_api.hip).  The f32 and f64 16x16x4 MFMA differ in the C/D row map.
// ---------------------------------------------------------------------------------------
typedef double lu_v4d __attribute__((ext_vector_type(4)));
typedef float lu_v4f __attribute__((ext_vector_type(4)));
__host__ __device__ inline cplxf cdiv(cplxf a, cplxf b) {
    float d = b.x * b.x + b.y * b.y;
    return cmakef((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}
template <typename T> struct lu_el;
template <> struct lu_el<cplx> {
    typedef lu_v4d v4;
    __host__ __device__ static cplx mk(double a, double b) { return cmake(a, b); }
    __device__ static int mrow(int lk, int r) { return lk + 4 * r; }
    __device__ static v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};
template <> struct lu_el<cplxf> {
    typedef lu_v4f v4;
    __host__ __device__ static cplxf mk(double a, double b) { return cmakef((float)a, (float)b); }
    __device__ static int mrow(int lk, int r) { return 4 * lk + r; }
    __device__ static v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};
#define LU_MK(a, b) lu_el<T>::mk((a), (b))
// Geometry of a factor the LU / substitution kernels work on.  Dense factor: ld = n = N, the inverted diagonal blocks sit
// behind the N x N matrix.  Band factor (fh_wband_*, below): the SAME kernels run on general band storage viewed as a
// column-major matrix with leading dimension ldab - 1 (element (i, j) of the band at base + i + j (ldab - 1)), n = matrix
// order for the row / column bounds, and the inverses behind the band array.
struct lu_geom {
    int ld;             // leading dimension of the column-major view
    int n;              // rows = columns of the matrix (bounds)
    size_t inv32;       // offset (elements, from the matrix pointer) of the LU_NB-block inverses
    size_t inv128;      // offset of the SOLVE_KB-block inverses
};
template <typename VT, bool BIDENT, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_form_shifted(const VT* __restrict__ A, const VT* __restrict__ B,
                                                            T* const* LUs, const cplx* z, int N) {
    T* S = LUs[blockIdx.y];
    const cplx zz = z[blockIdx.y];
    const size_t total = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        cplx v;
        if constexpr (sizeof(VT) == sizeof(cplx)) v = cmake(-A[e].x, -A[e].y); else v = cmake(-A[e], 0.0);
        if (BIDENT) {
            if (e % N == e / N) v = cadd(v, zz);
        } else {
            cplx b;
            if constexpr (sizeof(VT) == sizeof(cplx)) b = cmake(B[e].x, B[e].y); else b = cmake(B[e], 0.0);
            cfma(v, zz, b);
        }
        S[e] = cvt<T>(v);
    }
}

// Panel factorisation: one workgroup per matrix walks the nb panel columns; pivot rule is
// LAPACK's IZAMAX (max |re|+|im|, lowest index on ties).
template <typename T>
__global__ __launch_bounds__(LU_PANEL_THREADS) void k_lu_panel(T* const* LUs, int* const* pivs, int N, int k0,
                                                                int nb, int* info) {
    T* A = LUs[blockIdx.x];
    int* piv = pivs[blockIdx.x];
    __shared__ double smax[LU_PANEL_THREADS];
    __shared__ int sidx[LU_PANEL_THREADS];
    __shared__ int sp;
    const int t = threadIdx.x;
    for (int j = 0; j < nb; ++j) {
        const int jj = k0 + j;
        double best = -1.0;
        int bi = jj;
        for (int i = jj + t; i < N; i += LU_PANEL_THREADS) {
            T v = A[(size_t)jj * N + i];
            double m = fabs(v.x) + fabs(v.y);
            if (m > best) { best = m; bi = i; }
        }
        smax[t] = best; sidx[t] = bi;
        __syncthreads();
        for (int s = LU_PANEL_THREADS / 2; s > 0; s >>= 1) {
            if (t < s) {
                double o = smax[t + s]; int oi = sidx[t + s];
                if (o > smax[t] || (o == smax[t] && oi < sidx[t])) { smax[t] = o; sidx[t] = oi; }
            }
            __syncthreads();
        }
        if (t == 0) {
            sp = sidx[0];
            piv[jj] = sidx[0];
            if (!(smax[0] > 0.0) || !isfinite(smax[0])) { if (info[blockIdx.x] == 0) info[blockIdx.x] = jj + 1; }
        }
        __syncthreads();
        const int p = sp;
        if (t < nb && p != jj) {
            T u = A[(size_t)(k0 + t) * N + jj];
            A[(size_t)(k0 + t) * N + jj] = A[(size_t)(k0 + t) * N + p];
            A[(size_t)(k0 + t) * N + p] = u;
        }
        __syncthreads();
        const T pv = A[(size_t)jj * N + jj];
        const bool singular = (pv.x == 0.0 && pv.y == 0.0);
        const T inv = singular ? LU_MK(0, 0) : cdiv(LU_MK(1, 0), pv);
        for (int i = jj + 1 + t; i < N; i += LU_PANEL_THREADS) A[(size_t)jj * N + i] = cmul(A[(size_t)jj * N + i], inv);
        __syncthreads();
        const int ncols = k0 + nb - 1 - jj, nrows = N - 1 - jj;
        for (int e = t; e < nrows * ncols; e += LU_PANEL_THREADS) {
            int i = jj + 1 + e % nrows, c = jj + 1 + e / nrows;
            T l = A[(size_t)jj * N + i], u = A[(size_t)c * N + jj];
            T v = A[(size_t)c * N + i];
            A[(size_t)c * N + i] = csub(v, cmul(l, u));
        }
        __syncthreads();
    }
}

// broadcast of one lane's value with a wave-uniform lane index: v_readlane (scalar path, a few cycles) instead of the
// ds_bpermute a generic __shfl costs (LDS round trip, on the dependent chain of the substitutions below)
__device__ inline double lu_readlane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ inline float lu_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// Wave-wide pivot search: largest `best`, lowest row index `bi` among equals (the IZAMAX rule), result in every lane.
// Inside a 16-lane row by DPP (quad xor 1, quad xor 2, half-row mirror, row mirror: after each step both partners hold
// the max of their union), across the four rows by v_readlane.  The __shfl_xor butterfly this replaces cost 18
// ds_bpermute round trips per column on the panel's critical path (13 % of the panel kernel, measured by knock-out).
template <int CTRL>
__device__ inline double lu_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ inline void lu_wave_argmax(double& best, int& bi) {
    double m = best;
    m = fmax(m, lu_dpp_f64<0xB1>(m));      // quad_perm [1,0,3,2]
    m = fmax(m, lu_dpp_f64<0x4E>(m));      // quad_perm [2,3,0,1]
    m = fmax(m, lu_dpp_f64<0x141>(m));     // row_half_mirror
    m = fmax(m, lu_dpp_f64<0x140>(m));     // row_mirror
    m = fmax(fmax(lu_readlane(m, 0), lu_readlane(m, 16)), fmax(lu_readlane(m, 32), lu_readlane(m, 48)));
    int c = (best == m) ? bi : 0x7fffffff;
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0xB1, 0xF, 0xF, false));
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0x4E, 0xF, 0xF, false));
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0x141, 0xF, 0xF, false));
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0x140, 0xF, 0xF, false));
    c = min(min(__builtin_amdgcn_readlane(c, 0), __builtin_amdgcn_readlane(c, 16)),
            min(__builtin_amdgcn_readlane(c, 32), __builtin_amdgcn_readlane(c, 48)));
    best = m;
    bi = c;
}

// Register-resident panel factorisation.  One workgroup (1024 threads) per matrix; thread t
// owns rows k0 + t + 1024 r (r < R) of the panel.  The nb panel columns are processed
// left-looking in sub-panels of W columns (R*W = 16 complex values = 64 VGPRs per thread):
//   1. U part of the sub-panel: forward substitution with the unit-lower block of the previous
//      panel columns, in LDS (W threads);
//   2. owned rows of the sub-panel are loaded into registers and updated with the previous
//      columns (one coalesced read of each previous column, U part broadcast from LDS);
//   3. the W columns are eliminated in registers: pivot search = per-thread max, wave shuffle
//      reduction, 16-entry LDS reduction; pivot row and row jj exchanged through LDS; rank-1
//      update in registers.  Two workgroup barriers per column and no global-memory traffic
//      except the row interchange of the other panel columns.
// Pivot rule as k_lu_panel (LAPACK IZAMAX).  Each panel entry is read from and written to
// global memory once per sub-panel it participates in, instead of once per column.
// NT: threads per workgroup.  1024 for the dense and band factorisations (one matrix per CU, every wave slot taken); the
// multifrontal fronts of <= 256 / 512 rows come in batches of thousands and take 256 / 512 threads, so that four / two of
// them share a CU instead of one with three quarters of its lanes idle.
template <int R, int W, typename T, int NT = 1024>
__global__ __launch_bounds__(NT) void k_lu_panel_reg(T* const* LUs, int* const* pivs, lu_geom g, int nr, int k0,
                                                                    int nb, int* info, int pl) {
    // pl: pivot limit -- rows >= pl are eliminated but never chosen as a pivot (pl = nr: LAPACK's partial pivoting; the
    // multifrontal fronts pass the size of their fully-summed block)
    T* A = LUs[blockIdx.x];
    const int N = g.ld;                          // leading dimension; rows of the panel: [k0, nr)
    int* piv = pivs[blockIdx.x];
    __shared__ double wmax[NT / 64];
    __shared__ int widx[NT / 64];
    __shared__ T rowA[W], rowB[W];
    __shared__ T Lsm[LU_NB][LU_NB + 1];
    __shared__ T Us[LU_NB][W];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int kend = k0 + nb;
    int rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = k0 + t + NT * r;

    for (int c0 = k0; c0 < kend; c0 += W) {
        const int pc = c0 - k0;                      // previous panel columns
        const int wact = min(W, kend - c0);
        // ---- 1. U part: Us = L11^-1 A[k0:c0, c0:c0+wact]
        if (pc > 0) {
            for (int e = t; e < pc * pc; e += NT) {
                int i = e % pc, j = e / pc;
                Lsm[i][j] = A[(size_t)(k0 + j) * N + k0 + i];
            }
            for (int e = t; e < pc * wact; e += NT) {
                int i = e % pc, w = e / pc;
                Us[i][w] = A[(size_t)(c0 + w) * N + k0 + i];
            }
            __syncthreads();
            // forward substitution with one wave per sub-panel column (W <= 16 waves): lane = row of the block, the
            // solved component is broadcast by shuffle and every later row updates itself -- pc short steps instead
            // of pc^2/2 dependent operations of a single thread
            for (int wc = wave; wc < wact; wc += NT / 64) {
                T x = lane < pc ? Us[lane][wc] : LU_MK(0, 0);
                for (int i = 0; i + 1 < pc; ++i) {
                    const T xi = LU_MK(lu_readlane(x.x, i), lu_readlane(x.y, i));
                    if (lane > i && lane < pc) x = csub(x, cmul(Lsm[lane][i], xi));
                }
                if (lane < pc) {
                    Us[lane][wc] = x;
                    A[(size_t)(c0 + wc) * N + k0 + lane] = x;
                }
            }
            __syncthreads();
        }
        // ---- 2. load owned rows (>= c0) and apply the previous columns
        T a[R][W];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int w = 0; w < W; ++w)
                a[r][w] = (rows[r] >= c0 && rows[r] < nr && w < wact) ? A[(size_t)(c0 + w) * N + rows[r]] : LU_MK(0, 0);
        // The loop is bound by the latency of these streaming loads of L, not by their bytes.  complex64 has the
        // registers for two previous columns per step (pc is a multiple of W, so even); complex128 at 1024 threads
        // (128 VGPRs) does not -- the two-column form spilled -- and keeps one column per step.
        if constexpr (sizeof(T) == sizeof(cplxf) && R <= 4) {
            for (int p = 0; p < pc; p += 2) {
                T l0[R], l1[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool own = rows[r] >= c0 && rows[r] < nr;
                    l0[r] = own ? A[(size_t)(k0 + p) * N + rows[r]] : LU_MK(0, 0);
                    l1[r] = own ? A[(size_t)(k0 + p + 1) * N + rows[r]] : LU_MK(0, 0);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        a[r][w] = csub(a[r][w], cmul(l0[r], Us[p][w]));
                        a[r][w] = csub(a[r][w], cmul(l1[r], Us[p + 1][w]));
                    }
                }
            }
        } else if constexpr (R == 1) {
            // one row per thread: two previous columns per step, their loads in flight together (pc is a multiple of W = 16;
            // four at a time spill 18 VGPRs).
            // Beside the trailing product of the look-ahead a single dependent load takes microseconds; same operations in the
            // same order on every element.
            const bool own = rows[0] >= c0 && rows[0] < nr;
            for (int p = 0; p < pc; p += 2) {
                T l[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) l[q] = own ? A[(size_t)(k0 + p + q) * N + rows[0]] : LU_MK(0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int w = 0; w < W; ++w) a[0][w] = csub(a[0][w], cmul(l[q], Us[p + q][w]));
            }
        } else {
            for (int p = 0; p < pc; ++p) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (rows[r] >= c0 && rows[r] < nr) {
                        const T l = A[(size_t)(k0 + p) * N + rows[r]];
#pragma unroll
                        for (int w = 0; w < W; ++w) a[r][w] = csub(a[r][w], cmul(l, Us[p][w]));
                    }
                }
            }
        }
        // ---- 3. eliminate the sub-panel columns in registers
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int jj = c0 + j;
            if (jj < kend) {                          // uniform
                double best = -1.0;
                int bi = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (rows[r] >= jj && rows[r] < pl) {
                        const double m = fabs(a[r][j].x) + fabs(a[r][j].y);
                        if (m > best) { best = m; bi = rows[r]; }
                    }
                }
                lu_wave_argmax(best, bi);
                if (lane == 0) { wmax[wave] = best; widx[wave] = bi; }
                __syncthreads();
                best = wmax[0]; bi = widx[0];
#pragma unroll
                for (int q = 1; q < NT / 64; ++q) {
                    const double o = wmax[q];
                    const int oi = widx[q];
                    if (o > best || (o == best && oi < bi)) { best = o; bi = oi; }
                }
                const int p = (bi == 0x7fffffff) ? jj : bi;
                if (t == 0) {
                    piv[jj] = p;
                    if (!(best > 0.0) || !isfinite(best)) { if (info[blockIdx.x] == 0) info[blockIdx.x] = jj + 1; }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (rows[r] == jj) {
#pragma unroll
                        for (int w = 0; w < W; ++w) rowA[w] = a[r][w];
                    }
                    if (rows[r] == p) {
#pragma unroll
                        for (int w = 0; w < W; ++w) rowB[w] = a[r][w];
                    }
                }
                // the panel columns outside this sub-panel: interchange in global memory
                if (p != jj && t < nb && (k0 + t < c0 || k0 + t >= c0 + W)) {
                    T u = A[(size_t)(k0 + t) * N + jj];
                    A[(size_t)(k0 + t) * N + jj] = A[(size_t)(k0 + t) * N + p];
                    A[(size_t)(k0 + t) * N + p] = u;
                }
                __syncthreads();
                const T pv = rowB[j];
                const bool singular = (pv.x == 0.0 && pv.y == 0.0);
                const T inv = singular ? LU_MK(0, 0) : cdiv(LU_MK(1, 0), pv);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (p != jj) {
                        if (rows[r] == jj) {
#pragma unroll
                            for (int w = 0; w < W; ++w) a[r][w] = rowB[w];
                        } else if (rows[r] == p) {
#pragma unroll
                            for (int w = 0; w < W; ++w) a[r][w] = rowA[w];
                        }
                    }
                    if (rows[r] > jj && rows[r] < nr) {
                        const T l = cmul(a[r][j], inv);
                        a[r][j] = l;
#pragma unroll
                        for (int w = j + 1; w < W; ++w) a[r][w] = csub(a[r][w], cmul(l, rowB[w]));
                    }
                }
            }
        }
        // ---- store rows >= c0 of the sub-panel
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int w = 0; w < W; ++w)
                if (rows[r] >= c0 && rows[r] < nr && w < wact) A[(size_t)(c0 + w) * N + rows[r]] = a[r][w];
        __syncthreads();
    }
    // ---- inverse of the unit-lower diagonal block, for the U block row (k_lu_trsm_mul): column c of L11^-1 by forward
    // substitution on e_c, one wave per column, lane = row, the solved component broadcast by shuffle
    if (nb == LU_NB) {
        for (int e = t; e < LU_NB * LU_NB; e += NT) {
            const int i = e % LU_NB, j = e / LU_NB;
            Lsm[i][j] = A[(size_t)(k0 + j) * N + k0 + i];
        }
        __syncthreads();
        T* inv = A + g.inv32 + (size_t)(k0 / LU_NB) * 2 * LU_NB * LU_NB;
        for (int c = wave; c < LU_NB; c += NT / 64) {
            T x = LU_MK(lane == c ? 1.0 : 0.0, 0.0);
            for (int i = c; i + 1 < LU_NB; ++i) {
                const T xi = LU_MK(lu_readlane(x.x, i), lu_readlane(x.y, i));
                if (lane > i && lane < LU_NB) x = csub(x, cmul(Lsm[lane][i], xi));
            }
            if (lane < LU_NB) inv[c * LU_NB + lane] = x;
        }
    }
}

// apply the row interchanges piv[p0 .. p0+np) to the columns [a0,a1) and [b0,b1)
template <typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_lu_laswp(T* const* LUs, int* const* pivs, int N, int p0, int np,
                                                        int a0, int a1, int b0, int b1) {
    T* A = LUs[blockIdx.y];
    const int* piv = pivs[blockIdx.y];
    const int id = blockIdx.x * FH_BLOCK + threadIdx.x;
    const int na = a1 - a0;
    const int c = id < na ? a0 + id : b0 + (id - na);
    if (c >= b1) return;
    for (int j = 0; j < np; ++j) {
        int p = piv[p0 + j];
        if (p != p0 + j) {
            T u = A[(size_t)c * N + p0 + j];
            A[(size_t)c * N + p0 + j] = A[(size_t)c * N + p];
            A[(size_t)c * N + p] = u;
        }
    }
}

// U block row: A[k0:k0+NB, c] = L11^{-1} A[k0:k0+NB, c] for columns c in [c0,c1); one thread per column
template <int NB, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_lu_trsm(T* const* LUs, int N, int k0, int c0, int c1) {
    T* A = LUs[blockIdx.y];
    __shared__ T L[NB][NB + 1];
    for (int e = threadIdx.x; e < NB * NB; e += FH_BLOCK) {
        int i = e % NB, j = e / NB;
        L[i][j] = A[(size_t)(k0 + j) * N + k0 + i];
    }
    __syncthreads();
    const int c = c0 + blockIdx.x * FH_BLOCK + threadIdx.x;
    if (c >= c1) return;
    T x[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) x[i] = A[(size_t)c * N + k0 + i];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int i = j + 1; i < NB; ++i) x[i] = csub(x[i], cmul(L[i][j], x[j]));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) A[(size_t)c * N + k0 + i] = x[i];
}

// U block row as a product with the inverse the panel kernel left behind the factor:
// A[k0:k0+NB, c] = L11^-1 A[k0:k0+NB, c].  One thread per ENTRY (NB rows x 8 columns per workgroup): 32 independent
// multiply-adds instead of the substitution's 31 dependent steps per thread, coalesced 512 B column segments.
template <int NB, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_lu_trsm_mul(T* const* LUs, lu_geom g, int k0, int c0, int c1) {
    static_assert(FH_BLOCK % NB == 0, "rows x columns tiling");
    constexpr int CW = FH_BLOCK / NB;
    T* A = LUs[blockIdx.y];
    const int N = g.ld;
    const T* inv = A + g.inv32 + (size_t)(k0 / NB) * 2 * NB * NB;
    __shared__ T Li[NB * NB];          // column-major: Li[j * NB + i], lanes read consecutive i
    __shared__ T a[CW][NB];
    const int t = threadIdx.x, i = t % NB, cc = t / NB;
    for (int e = t; e < NB * NB; e += FH_BLOCK) Li[e] = inv[e];
    const int c = c0 + blockIdx.x * CW + cc;
    if (c < c1) a[cc][i] = A[(size_t)c * N + k0 + i];
    __syncthreads();
    if (c >= c1) return;
    T s = LU_MK(0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) cfma(s, Li[j * NB + i], a[cc][j]);
    A[(size_t)c * N + k0 + i] = s;
}

// A[r0:r1, c0:c1] -= A[r0:r1, k0:k0+kd] * A[k0:k0+kd, c0:c1]  (kd a multiple of KC = 32) on
// v_mfma_f64_16x16x4_f64: 64x64 tile per workgroup, four waves as 2x2 of 32x32, each wave
// 2x2 MFMA tiles; the k range is walked in chunks of KC through LDS.  The product is formed
// TRANSPOSED,   D[c][i] = sum_k U[k][c] * L[i][k]      (A operand = U^T, B operand = L^T)
// so that a lane's results (row = (l>>4)+4r -> c, col = l&15 -> i) are 16 consecutive rows of
// one column of the column-major matrix: 256 B contiguous per 16-lane group.
// Complex product with two accumulators per tile: Re += ur*lr + (-ui)*li, Im += ur*li + ui*lr.
// The two-level factorisation calls it with kd = 32 inside an outer block column and kd = 128
// for the trailing matrix, which is then read and written once per 128 eliminated columns.
template <int KC, typename T>
__global__ __launch_bounds__(FH_BLOCK, 2) void k_lu_gemm(T* const* LUs, int N, int k0, int kd, int r0, int r1, int c0,
                                                       int c1, int TR, int TC, int compact = 0) {
    T* A = LUs[blockIdx.y];
    __shared__ T Ls[KC][64];
    __shared__ T Us[KC][64];
    const int t = threadIdx.x;
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs, so XCD x takes the
    // 8x8 super-tiles x, x+8, ... and walks one super-tile with 64 consecutive local slots:
    // its L2 then holds the 8 L row tiles and <= 8 U column tiles (2 MB at k = 128) being reused
    int tr, tc;
    if (compact) {                                             // (the multifrontal fronts: small products in large batches, see k_lu_gemm_direct)
        tr = blockIdx.x % TR; tc = blockIdx.x / TR;
    } else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int SR = (TR + 7) >> 3;
        const int sw = min(8, TC);                             // super-tile: 8 row tiles x sw column tiles
        const int st = (slot / (8 * sw)) * 8 + xcd, within = slot % (8 * sw);
        tr = (st % SR) * 8 + (within & 7); tc = (st / SR) * sw + (within >> 3);
    }
    if (tr >= TR || tc >= TC) return;
    const int i0 = r0 + tr * 64, cc0 = c0 + tc * 64;
    const int lane = t & 63, wave = t >> 6;
    const int wi = (wave & 1) * 32, wc = (wave >> 1) * 32;     // wave's 32x32 sub-tile
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int PF = KC * 64 / FH_BLOCK;                     // panel entries per thread and chunk
    T pl[PF], pu[PF];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int e = t + q * FH_BLOCK;
            const int ii = e % 64, kl = e / 64;
            pl[q] = (i0 + ii < r1) ? A[(size_t)(k0 + kc + kl) * N + i0 + ii] : LU_MK(0, 0);
            const int ku = e % KC, cc = e / KC;
            pu[q] = (cc0 + cc < c1) ? A[(size_t)(cc0 + cc) * N + k0 + kc + ku] : LU_MK(0, 0);
        }
    };
    fetch(0);
    typename lu_el<T>::v4 re[2][2], im[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { re[a][b] = (typename lu_el<T>::v4){0, 0, 0, 0}; im[a][b] = re[a][b]; }
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int e = t + q * FH_BLOCK;
            Ls[e / 64][e % 64] = pl[q];
            Us[e % KC][(e / KC) ^ ((e % KC) & 15)] = pu[q];   // XOR swizzle: the transposed store hits 16 bank groups, not one
        }
    };
    auto mma = [&]() {
#pragma unroll 2
        for (int kk = 0; kk < KC; kk += 4) {
            T u[2], l[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) u[a] = Us[kk + lk][(wc + 16 * a + lr) ^ ((kk + lk) & 15)];
#pragma unroll
            for (int b = 0; b < 2; ++b) l[b] = Ls[kk + lk][wi + 16 * b + lr];
            // four sweeps over the 2x2 tiles: dependent MFMAs on one accumulator are 8 issues apart
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    re[a][b] = lu_el<T>::mfma(u[a].x, l[b].x, re[a][b]);
                    im[a][b] = lu_el<T>::mfma(u[a].x, l[b].y, im[a][b]);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    re[a][b] = lu_el<T>::mfma(-u[a].y, l[b].y, re[a][b]);
                    im[a][b] = lu_el<T>::mfma(u[a].y, l[b].x, im[a][b]);
                }
        }
    };
    int kc = 0;
    for (; kc + KC < kd; kc += KC) {
        if (kc) __syncthreads();
        stage();
        __syncthreads();
        fetch(kc + KC);                            // next chunk's loads fly under this chunk's MFMAs
        mma();
    }
    // last chunk: the panel registers are free, this lane's 16 matrix entries take their place
    if (kc) __syncthreads();
    stage();
    __syncthreads();
    T cv[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = cc0 + wc + 16 * a + lu_el<T>::mrow(lk, r);
                const int i = i0 + wi + 16 * b + lr;
                cv[a][b][r] = (i < r1 && c < c1) ? A[(size_t)c * N + i] : LU_MK(0, 0);
            }
    mma();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = cc0 + wc + 16 * a + lu_el<T>::mrow(lk, r);
                const int i = i0 + wi + 16 * b + lr;
                if (i < r1 && c < c1) {
                    T v = cv[a][b][r];
                    v.x -= re[a][b][r];
                    v.y -= im[a][b][r];
                    A[(size_t)c * N + i] = v;
                }
            }
}

// Second form of the trailing update (default): the L panel is no longer staged through LDS.  A wave owns 16
// rows of the 64 x 64 tile and all four 16-column tiles: its L operand (16 consecutive rows of one factor column,
// one 256 B segment per 16 lanes) comes straight from global memory, once per k-step, and is reused by the four
// column tiles; only the U block -- whose contiguous direction (k) is the wrong one for an MFMA operand -- goes
// through LDS (transposed, XOR-swizzled store).  Same transposed product and super-tile order as k_lu_gemm.
// M3 (complex128 only): the complex product with THREE real MFMA products per k-step instead of four (Karatsuba / "3M"):
//     P1 += ur lr,  P2 += ui li,  P3 += (ur + ui)(lr + li);      Re = P1 - P2,  Im = P3 - P1 - P2
// on three accumulator tiles per column tile; the operand sums cost five fp64 adds per k-step and wave against four
// saved MFMAs.  The imaginary part of a product is then formed by cancellation (componentwise error eps (|ur| + |ui|)
// (|lr| + |li|) instead of eps (|ur li| + |ui lr|)): normwise the update is as accurate as before -- the LU tests against
// LAPACK hold at their tolerances -- which is the guarantee ZGEMM3M gives.  FH_LU_3M=0 selects the four-product form.
template <int KC, typename T, bool M3>
__global__ __launch_bounds__(FH_BLOCK, 2) void k_lu_gemm_direct(T* const* LUs, int N, int k0, int kd, int r0, int r1, int c0,
                                                              int c1, int TR, int TC, int compact = 0) {
    T* A = LUs[blockIdx.y];
    __shared__ T Us[KC][64];
    const int t = threadIdx.x;
    int tr, tc;
    if (compact) {                              // compact grid (small products in large batches: the multifrontal fronts) -- no idle workgroups
        tr = blockIdx.x % TR; tc = blockIdx.x / TR;
    } else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int SR = (TR + 7) >> 3;
        const int sw = min(8, TC);
        const int st = (slot / (8 * sw)) * 8 + xcd, within = slot % (8 * sw);
        tr = (st % SR) * 8 + (within & 7); tc = (st / SR) * sw + (within >> 3);
    }
    if (tr >= TR || tc >= TC) return;
    const int i0 = r0 + tr * 64, cc0 = c0 + tc * 64;
    const int lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int irow = i0 + 16 * wave + lr;
    constexpr int PF = KC * 64 / FH_BLOCK;
    T pu[PF], ln[KC / 4];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int e = t + q * FH_BLOCK;
            const int ku = e % KC, cc = e / KC;
            pu[q] = (cc0 + cc < c1) ? A[(size_t)(cc0 + cc) * N + k0 + kc + ku] : LU_MK(0, 0);
        }
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) ln[s] = (irow < r1) ? A[(size_t)(k0 + kc + 4 * s + lk) * N + irow] : LU_MK(0, 0);
    };
    fetch(0);
    typename lu_el<T>::v4 re[4], im[4], p3[M3 ? 4 : 1];
#pragma unroll
    for (int a = 0; a < 4; ++a) { re[a] = (typename lu_el<T>::v4){0, 0, 0, 0}; im[a] = re[a]; }
#pragma unroll
    for (int a = 0; a < (M3 ? 4 : 1); ++a) p3[a] = (typename lu_el<T>::v4){0, 0, 0, 0};
    T lc[KC / 4];
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int e = t + q * FH_BLOCK;
            Us[e % KC][(e / KC) ^ ((e % KC) & 15)] = pu[q];
        }
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) lc[s] = ln[s];
    };
    auto mma = [&]() {
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            T u[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) u[a] = Us[4 * s + lk][(16 * a + lr) ^ ((4 * s + lk) & 15)];
            if (M3) {
                // re[] holds P1, im[] holds P2, p3[] holds P3 until the epilogue
                const auto lsum = lc[s].x + lc[s].y;
#pragma unroll
                for (int a = 0; a < 4; ++a) re[a] = lu_el<T>::mfma(u[a].x, lc[s].x, re[a]);
#pragma unroll
                for (int a = 0; a < 4; ++a) im[a] = lu_el<T>::mfma(u[a].y, lc[s].y, im[a]);
#pragma unroll
                for (int a = 0; a < 4; ++a) p3[a] = lu_el<T>::mfma(u[a].x + u[a].y, lsum, p3[a]);
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a) re[a] = lu_el<T>::mfma(u[a].x, lc[s].x, re[a]);
#pragma unroll
                for (int a = 0; a < 4; ++a) im[a] = lu_el<T>::mfma(u[a].x, lc[s].y, im[a]);
#pragma unroll
                for (int a = 0; a < 4; ++a) re[a] = lu_el<T>::mfma(-u[a].y, lc[s].y, re[a]);
#pragma unroll
                for (int a = 0; a < 4; ++a) im[a] = lu_el<T>::mfma(u[a].y, lc[s].x, im[a]);
            }
        }
    };
    int kc = 0;
    for (; kc + KC < kd; kc += KC) {
        if (kc) __syncthreads();
        stage();
        __syncthreads();
        fetch(kc + KC);
        mma();
    }
    if (kc) __syncthreads();
    stage();
    __syncthreads();
    T cv[4][4];
    auto load_c = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = cc0 + 16 * a + lu_el<T>::mrow(lk, r);
                cv[a][r] = (irow < r1 && c < c1) ? A[(size_t)c * N + irow] : LU_MK(0, 0);
            }
    };
    // (four-product form: the C tile is loaded under the last chunk's MFMAs; with the third accumulator set there are no
    //  registers left for that -- 19 VGPRs spilled -- so the three-product form loads it after the last chunk)
    if (!M3) load_c();
    mma();
    if (M3) load_c();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = cc0 + 16 * a + lu_el<T>::mrow(lk, r);
            if (irow < r1 && c < c1) {
                T v = cv[a][r];
                if (M3) {
                    v.x -= re[a][r] - im[a][r];                          // P1 - P2
                    v.y -= p3[a][r] - re[a][r] - im[a][r];               // P3 - P1 - P2
                } else {
                    v.x -= re[a][r];
                    v.y -= im[a][r];
                }
                A[(size_t)c * N + irow] = v;
            }
        }
}

// ---- solve ------------------------------------------------------------------------------
// row permutation of the whole factorisation from the LAPACK-style pivot list, built once per
// factorisation in LDS (N <= 16384) and cached behind the pivots: perm = pivs[q] + N
__global__ __launch_bounds__(FH_BLOCK) void k_build_perm(int* const* pivs, int N) {
    extern __shared__ int sperm[];
    const int* piv = pivs[blockIdx.x];
    int* perm = pivs[blockIdx.x] + N;
    for (int i = threadIdx.x; i < N; i += FH_BLOCK) sperm[i] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < N; ++i) {
            int p = piv[i];
            if (p != i) { int u = sperm[i]; sperm[i] = sperm[p]; sperm[p] = u; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += FH_BLOCK) perm[i] = sperm[i];
}

// same, in global memory, for N beyond the LDS capacity
__global__ void k_build_perm_global(int* const* pivs, int N) {
    if (threadIdx.x != 0) return;
    const int* piv = pivs[blockIdx.x];
    int* perm = pivs[blockIdx.x] + N;
    for (int i = 0; i < N; ++i) perm[i] = i;
    for (int i = 0; i < N; ++i) {
        int p = piv[i];
        if (p != i) { int u = perm[i]; perm[i] = perm[p]; perm[p] = u; }
    }
}

// inverses of the NB x NB diagonal blocks of L (unit lower) and U, stored behind the factor:
// inv = LU + N*N + (2*block + upper)*NB*NB, column-major, identity-padded in a short last
// block.  The block triangular solves then become products (k_solve_step), which removes the
// nb-step sequential substitution from the critical path of every block step.
template <int NB, typename T>
__global__ __launch_bounds__(64) void k_lu_invert_diag(T* const* LUs, lu_geom g) {
    T* A = LUs[blockIdx.y];
    const int N = g.ld;
    const int k0 = blockIdx.x * NB;
    const int nb = min(NB, g.n - k0);
    T* inv = A + g.inv32 + (size_t)blockIdx.x * 2 * NB * NB;
    __shared__ T Tb[NB][NB + 1];
    __shared__ T Li[NB][NB + 1];
    __shared__ T Ui[NB][NB + 1];
    const int t = threadIdx.x;
    for (int e = t; e < NB * NB; e += 64) {
        int i = e % NB, j = e / NB;
        Tb[i][j] = (i < nb && j < nb) ? A[(size_t)(k0 + j) * N + k0 + i] : LU_MK(i == j ? 1.0 : 0.0, 0.0);
        Li[i][j] = LU_MK(0, 0);
        Ui[i][j] = LU_MK(0, 0);
    }
    __syncthreads();
    if (t < NB) {                       // column t of L^-1
        const int c = t;
        Li[c][c] = LU_MK(1, 0);
        for (int i = c + 1; i < NB; ++i) {
            T s = LU_MK(0, 0);
            for (int j = c; j < i; ++j) cfma(s, Tb[i][j], Li[j][c]);
            Li[i][c] = LU_MK(-s.x, -s.y);
        }
    } else if (t < 2 * NB) {            // column c of U^-1
        const int c = t - NB;
        Ui[c][c] = cdiv(LU_MK(1, 0), Tb[c][c]);
        for (int i = c - 1; i >= 0; --i) {
            T s = LU_MK(0, 0);
            for (int j = i + 1; j <= c; ++j) cfma(s, Tb[i][j], Ui[j][c]);
            Ui[i][c] = cdiv(LU_MK(-s.x, -s.y), Tb[i][i]);
        }
    }
    __syncthreads();
    for (int e = t; e < NB * NB; e += 64) {
        int i = e % NB, j = e / NB;
        inv[e] = Li[i][j];
        inv[NB * NB + e] = Ui[i][j];
    }
}

// Y[node][i,:] = RHS[node][perm[i],:]   (rhs_stride = 0: one right-hand side panel shared by all nodes);
// the fp64 right-hand side is narrowed to the factor precision here
template <typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_gather_rows(const cplx* __restrict__ RHS, size_t rhs_stride, int* const* perms,
                                                           T* __restrict__ Y, size_t stride, int N, int ld) {
    const int* perm = perms[blockIdx.y];
    const cplx* Rn = RHS + (size_t)blockIdx.y * rhs_stride;
    T* Yn = Y + (size_t)blockIdx.y * stride;
    const size_t total = (size_t)N * ld;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        size_t i = e / ld, c = e % ld;
        Yn[e] = cvt<T>(Rn[(size_t)perm[i] * ld + c]);
    }
}

// complex64 solution panels back to the caller's fp64 panels
__global__ __launch_bounds__(FH_BLOCK) void k_widen_panels(const cplxf* __restrict__ src, cplx* __restrict__ dst, size_t stride,
                                                            size_t total) {
    const cplxf* s = src + (size_t)blockIdx.y * stride;
    cplx* d = dst + (size_t)blockIdx.y * stride;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) d[e] = to_d(s[e]);
}

// One block step of the triangular solves on row-major N x LD panels, one launch:
//   Z[slab]  = T^-1 * IN[slab]                       (T^-1 from k_lu_invert_diag)
//   IN[i,:] -= sum_j M[i, k0+j] Z[slab][j,:]         for i in [r0, r1)  (M = L below / U above)
// Every workgroup recomputes the 32 x LD slab product (tiny) into LDS; workgroup x = 0 also
// writes it to OUT.  The slab is read from IN and written to OUT, the updated rows are
// disjoint from the slab, so no workgroup reads what another writes.  Forward: IN = Y (permuted
// rhs), OUT = Z; backward: IN = Z, OUT = Y.  Both products run on v_mfma_f64_16x16x4_f64 with
// operands straight from global memory (A operand = 16 consecutive rows of one factor column,
// 256 B per 16 lanes); complex product with two accumulators: Re += ar*br + (-ai)*bi,
// Im += ar*bi + ai*br.  Wave w owns the 16-row band w of the 64-row tile and all LD columns.
template <int NB, int LD, bool UPPER, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_solve_step(T* const* LUs, T* IN, T* OUT, size_t stride, int N,
                                                          int k0, int r0, int r1, int rb, int cta) {
    static_assert(NB == 32, "tile mapping assumes NB == 32");
    const T* A = LUs[blockIdx.y];
    const T* inv = A + (size_t)N * N + ((size_t)(k0 / NB) * 2 + (UPPER ? 1 : 0)) * NB * NB;
    T* in = IN + (size_t)blockIdx.y * stride;
    T* out = OUT + (size_t)blockIdx.y * stride;
    __shared__ T Zs[NB][LD + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int CT = LD / 16;
    // ---- slab product
    // cta = column tiles that hold active right-hand sides (the rest of the panel is padding)
    for (int q = wave; q < 2 * cta; q += 4) {
        const int ti = q / cta, ta = q % cta;
        typename lu_el<T>::v4 re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < NB; kk += 4) {
            const T a = inv[(size_t)(kk + lk) * NB + 16 * ti + lr];
            const int row = k0 + kk + lk;
            const T b = row < N ? in[(size_t)row * LD + 16 * ta + lr] : LU_MK(0, 0);
            re = lu_el<T>::mfma(a.x, b.x, re);
            re = lu_el<T>::mfma(-a.y, b.y, re);
            im = lu_el<T>::mfma(a.x, b.y, im);
            im = lu_el<T>::mfma(a.y, b.x, im);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + lu_el<T>::mrow(lk, r), c = 16 * ta + lr;
            const T z = LU_MK(re[r], im[r]);
            Zs[i][c] = z;
            if (blockIdx.x == 0 && k0 + i < N) out[(size_t)(k0 + i) * LD + c] = z;
        }
    }
    __syncthreads();
    // ---- update: the workgroup owns 64*rb rows, wave w the 16-row bands w, w+4, ...
    for (int band = wave; band < 4 * rb; band += 4) {
        const int ib = r0 + (blockIdx.x * 4 * rb + band) * 16;
        if (ib >= r1) break;
        T am[NB / 4];
#pragma unroll
        for (int s = 0; s < NB / 4; ++s) {
            const int col = k0 + 4 * s + lk;
            am[s] = (ib + lr < r1 && col < N) ? A[(size_t)col * N + ib + lr] : LU_MK(0, 0);
        }
#pragma unroll
        for (int ta = 0; ta < CT; ++ta) {
            if (ta >= cta) break;
            T y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ib + lu_el<T>::mrow(lk, r);
                y[r] = i < r1 ? in[(size_t)i * LD + 16 * ta + lr] : LU_MK(0, 0);
            }
            typename lu_el<T>::v4 re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NB / 4; ++s) {
                const T b = Zs[4 * s + lk][16 * ta + lr];
                re = lu_el<T>::mfma(am[s].x, b.x, re);
                re = lu_el<T>::mfma(-am[s].y, b.y, re);
                im = lu_el<T>::mfma(am[s].x, b.y, im);
                im = lu_el<T>::mfma(am[s].y, b.x, im);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ib + lu_el<T>::mrow(lk, r);
                if (i < r1) in[(size_t)i * LD + 16 * ta + lr] = LU_MK(y[r].x - re[r], y[r].y - im[r]);
            }
        }
    }
}

// Two-level substitution (default): the solve walks SOLVE_KB = 128 columns per step instead of 32.
//   k_solve_diag    one workgroup per (column tile, node) keeps the 128 x 16 slab in LDS and runs the (up to)
//                   four 32-block steps inside it: z_i = T_i^-1 s_i, then s_j -= M_ji z_i for the blocks still
//                   to come -- the same products as k_solve_step, restricted to the 128 rows of the block;
//   k_solve_update  rows outside the block:  IN[i,:] -= sum_{k<kd} M[i, K0+k] Z[K0+k,:], a k = 128 product:
//                   one wave = a 16-row band x all active column tiles, the Z rows staged through LDS in
//                   16-deep chunks (split re/im, padded, register prefetch), factor operands from global.
// Against the 32-wide steps this rewrites the right-hand-side rows 4x less often and needs 2 launches per
// 128 columns instead of 4.
#define SOLVE_KB 128
#define SOLVE_KC 16

// IDENT: the right-hand side is the identity (grid = 8 column tiles x 128-blocks x nodes) and the result, the
// inverse of the 128 x 128 diagonal block, goes behind the 32-block inverses (column-major, identity-padded):
//   inv128 = LU + N*N + nblk32*2*32*32 + (2*block128 + upper)*128*128.
// With it the per-step diagonal solve of the substitution is one product (k_solve_diag_inv).
template <typename T>
__host__ __device__ inline size_t lu_inv128_offset(int N) {
    return (size_t)N * N + (size_t)((N + LU_NB - 1) / LU_NB) * 2 * LU_NB * LU_NB;
}

template <typename T>
static inline lu_geom lu_dense_geom(int N) { return lu_geom{N, N, (size_t)N * N, lu_inv128_offset<T>(N)}; }

template <int LD, bool UPPER, bool IDENT, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_solve_diag(T* const* LUs, T* IN, T* OUT, size_t stride, lu_geom g, int K0, int kb) {
    const int node = IDENT ? blockIdx.z : blockIdx.y;
    const T* A = LUs[node];
    const int N = g.ld, NBND = g.n;
    const T* invbase = A + g.inv32;
    if (IDENT) { K0 += blockIdx.y * SOLVE_KB; kb = min(SOLVE_KB / LU_NB, (NBND - K0 + LU_NB - 1) / LU_NB); }   // K0: first block of the launch
    const T* in = IN + (size_t)node * stride;
    T* out = OUT + (size_t)node * stride;
    const int ta = blockIdx.x;
    __shared__ T S[SOLVE_KB][17];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    for (int e = t; e < SOLVE_KB * 16; e += FH_BLOCK) {
        const int i = e >> 4, c = e & 15, row = K0 + i;
        if (IDENT) S[i][c] = LU_MK(i == 16 * ta + c ? 1.0 : 0.0, 0.0);
        else S[i][c] = (i < LU_NB * kb && row < NBND) ? in[(size_t)row * LD + 16 * ta + c] : LU_MK(0, 0);
    }
    __syncthreads();
    for (int step = 0; step < kb; ++step) {
        const int i = UPPER ? kb - 1 - step : step;
        const int k0 = K0 + LU_NB * i;
        const T* inv = invbase + ((size_t)(k0 / LU_NB) * 2 + (UPPER ? 1 : 0)) * LU_NB * LU_NB;
        typename lu_el<T>::v4 re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
        if (wave < 2) {
#pragma unroll
            for (int kk = 0; kk < LU_NB; kk += 4) {
                const T a = inv[(size_t)(kk + lk) * LU_NB + 16 * wave + lr];
                const T b = S[LU_NB * i + kk + lk][lr];
                re = lu_el<T>::mfma(a.x, b.x, re);
                re = lu_el<T>::mfma(-a.y, b.y, re);
                im = lu_el<T>::mfma(a.x, b.y, im);
                im = lu_el<T>::mfma(a.y, b.x, im);
            }
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[LU_NB * i + 16 * wave + lu_el<T>::mrow(lk, r)][lr] = LU_MK(re[r], im[r]);
        }
        __syncthreads();
        const int nj = UPPER ? i : kb - 1 - i;
        for (int q = wave; q < 2 * nj; q += 4) {
            const int j = UPPER ? q / 2 : i + 1 + q / 2;
            const int rbase = LU_NB * j + 16 * (q & 1);
            typename lu_el<T>::v4 ur = {0, 0, 0, 0}, ui = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < LU_NB; kk += 4) {
                const int grow = K0 + rbase + lr, gcol = k0 + kk + lk;
                const T a = (grow < NBND && gcol < NBND) ? A[(size_t)gcol * N + grow] : LU_MK(0, 0);
                const T b = S[LU_NB * i + kk + lk][lr];
                ur = lu_el<T>::mfma(a.x, b.x, ur);
                ur = lu_el<T>::mfma(-a.y, b.y, ur);
                ui = lu_el<T>::mfma(a.x, b.y, ui);
                ui = lu_el<T>::mfma(a.y, b.x, ui);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                T& d = S[rbase + lu_el<T>::mrow(lk, r)][lr];
                d = LU_MK(d.x - ur[r], d.y - ui[r]);
            }
        }
        __syncthreads();
    }
    if (IDENT) {
        T* inv128 = LUs[node] + g.inv128 + ((size_t)(K0 / SOLVE_KB) * 2 + (UPPER ? 1 : 0)) * SOLVE_KB * SOLVE_KB;
        for (int e = t; e < SOLVE_KB * 16; e += FH_BLOCK) {
            const int i = e & (SOLVE_KB - 1), c = e / SOLVE_KB;
            inv128[(size_t)(16 * ta + c) * SOLVE_KB + i] = S[i][c];
        }
        return;
    }
    for (int e = t; e < SOLVE_KB * 16; e += FH_BLOCK) {
        const int i = e >> 4, c = e & 15, row = K0 + i;
        if (i < LU_NB * kb && row < NBND) out[(size_t)row * LD + 16 * ta + c] = S[i][c];
    }
}

// z = T128^-1 s for one 128-row slab and one 16-column tile per workgroup.  The triangular structure bounds the
// contraction of output tile ti (lower: 16-blocks 0..ti; upper: ti..7); wave w takes the tiles w and 7 - w, nine
// 16-blocks together for every wave, and issues all nine operand loads before the slab is even in LDS: the
// kernel sits on the critical path of the substitution and is pure latency.
template <int LD, bool UPPER, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_solve_diag_inv(T* const* LUs, T* IN, T* OUT, size_t stride, lu_geom g, int K0, int kb) {
    constexpr int NT = SOLVE_KB / 16;
    const T* inv = LUs[blockIdx.y] + g.inv128 + ((size_t)(K0 / SOLVE_KB) * 2 + (UPPER ? 1 : 0)) * SOLVE_KB * SOLVE_KB;
    const T* in = IN + (size_t)blockIdx.y * stride;
    T* out = OUT + (size_t)blockIdx.y * stride;
    const int ta = blockIdx.x;
    __shared__ T S[SOLVE_KB][17];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int t0 = wave, t1 = NT - 1 - wave;
    const int n0 = UPPER ? NT - t0 : t0 + 1;
    const int blo0 = UPPER ? t0 : 0, blo1 = UPPER ? t1 : 0;
    T a[NT + 1][4];
#pragma unroll
    for (int q = 0; q <= NT; ++q) {
        const int tile = q < n0 ? t0 : t1, blk = q < n0 ? blo0 + q : blo1 + q - n0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) a[q][s4] = inv[(size_t)(16 * blk + 4 * s4 + lk) * SOLVE_KB + 16 * tile + lr];
    }
    for (int e = t; e < SOLVE_KB * 16; e += FH_BLOCK) {
        const int i = e >> 4, c = e & 15, row = K0 + i;
        S[i][c] = (i < LU_NB * kb && row < g.n) ? in[(size_t)row * LD + 16 * ta + c] : LU_MK(0, 0);
    }
    __syncthreads();
    typename lu_el<T>::v4 re0 = {0, 0, 0, 0}, im0 = {0, 0, 0, 0}, re1 = {0, 0, 0, 0}, im1 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q <= NT; ++q) {
        const int blk = q < n0 ? blo0 + q : blo1 + q - n0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const T b = S[16 * blk + 4 * s4 + lk][lr];
            if (q < n0) {
                re0 = lu_el<T>::mfma(a[q][s4].x, b.x, re0);
                re0 = lu_el<T>::mfma(-a[q][s4].y, b.y, re0);
                im0 = lu_el<T>::mfma(a[q][s4].x, b.y, im0);
                im0 = lu_el<T>::mfma(a[q][s4].y, b.x, im0);
            } else {
                re1 = lu_el<T>::mfma(a[q][s4].x, b.x, re1);
                re1 = lu_el<T>::mfma(-a[q][s4].y, b.y, re1);
                im1 = lu_el<T>::mfma(a[q][s4].x, b.y, im1);
                im1 = lu_el<T>::mfma(a[q][s4].y, b.x, im1);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i0 = 16 * t0 + lu_el<T>::mrow(lk, r), i1 = 16 * t1 + lu_el<T>::mrow(lk, r);
        if (i0 < LU_NB * kb && K0 + i0 < g.n) out[(size_t)(K0 + i0) * LD + 16 * ta + lr] = LU_MK(re0[r], im0[r]);
        if (i1 < LU_NB * kb && K0 + i1 < g.n) out[(size_t)(K0 + i1) * LD + 16 * ta + lr] = LU_MK(re1[r], im1[r]);
    }
}

// U block row of a whole 128-column block at once (band LU): A[K0:K0+128, c] = L11^-1 A[K0:K0+128, c] for the columns
// c in [c0, c1) of the column-major factor, L11^-1 the 128 x 128 inverse k_solve_diag<IDENT> leaves behind the factor.  The
// product is k_solve_diag_inv's (one workgroup per 16 columns, wave w the output tiles w and 7 - w); only the slab is read
// and written along columns instead of panel rows.  Replaces four L11^-1 products of 32 rows and three k = 32 row-block
// products per block column and side of the look-ahead.
template <typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_lu_trsm128(T* const* LUs, lu_geom g, int K0, int c0, int c1) {
    constexpr int NT = SOLVE_KB / 16;
    T* A = LUs[blockIdx.y];
    const T* inv = A + g.inv128 + ((size_t)(K0 / SOLVE_KB) * 2) * SOLVE_KB * SOLVE_KB;
    const int cbase = c0 + 16 * blockIdx.x;
    __shared__ T S[SOLVE_KB][17];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int t0 = wave, t1 = NT - 1 - wave;
    const int n0 = t0 + 1;
    T a[NT + 1][4];
#pragma unroll
    for (int q = 0; q <= NT; ++q) {
        const int tile = q < n0 ? t0 : t1, blk = q < n0 ? q : q - n0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) a[q][s4] = inv[(size_t)(16 * blk + 4 * s4 + lk) * SOLVE_KB + 16 * tile + lr];
    }
    for (int e = t; e < SOLVE_KB * 16; e += FH_BLOCK) {
        const int i = e & (SOLVE_KB - 1), c = e / SOLVE_KB;      // consecutive threads: consecutive rows of one column
        S[i][c] = (cbase + c < c1) ? A[(size_t)(cbase + c) * g.ld + K0 + i] : LU_MK(0, 0);
    }
    __syncthreads();
    typename lu_el<T>::v4 re0 = {0, 0, 0, 0}, im0 = {0, 0, 0, 0}, re1 = {0, 0, 0, 0}, im1 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q <= NT; ++q) {
        const int blk = q < n0 ? q : q - n0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const T b = S[16 * blk + 4 * s4 + lk][lr];
            if (q < n0) {
                re0 = lu_el<T>::mfma(a[q][s4].x, b.x, re0);
                re0 = lu_el<T>::mfma(-a[q][s4].y, b.y, re0);
                im0 = lu_el<T>::mfma(a[q][s4].x, b.y, im0);
                im0 = lu_el<T>::mfma(a[q][s4].y, b.x, im0);
            } else {
                re1 = lu_el<T>::mfma(a[q][s4].x, b.x, re1);
                re1 = lu_el<T>::mfma(-a[q][s4].y, b.y, re1);
                im1 = lu_el<T>::mfma(a[q][s4].x, b.y, im1);
                im1 = lu_el<T>::mfma(a[q][s4].y, b.x, im1);
            }
        }
    }
    if (cbase + lr < c1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i0 = 16 * t0 + lu_el<T>::mrow(lk, r), i1 = 16 * t1 + lu_el<T>::mrow(lk, r);
            A[(size_t)(cbase + lr) * g.ld + K0 + i0] = LU_MK(re0[r], im0[r]);
            A[(size_t)(cbase + lr) * g.ld + K0 + i1] = LU_MK(re1[r], im1[r]);
        }
    }
}

template <int LD, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_solve_update(T* const* LUs, T* IN, const T* ZS, size_t stride, lu_geom g, int K0,
                                                            int kd, int r0, int r1, int cta) {
    typedef decltype(T().x) ET;
    constexpr int KC = SOLVE_KC, CT = LD / 16, XPT = KC * LD / FH_BLOCK;
    __shared__ ET Xre[KC][LD + 16];
    __shared__ ET Xim[KC][LD + 16];
    const T* A = LUs[blockIdx.y];
    const int N = g.ld;
    T* in = IN + (size_t)blockIdx.y * stride;
    const T* zs = ZS + (size_t)blockIdx.y * stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int ib = r0 + (blockIdx.x * 4 + wave) * 16;
    const int irow = ib + lr;
    typename lu_el<T>::v4 re[CT], im[CT];
#pragma unroll
    for (int q = 0; q < CT; ++q) { re[q] = typename lu_el<T>::v4{0, 0, 0, 0}; im[q] = typename lu_el<T>::v4{0, 0, 0, 0}; }
    T xn[XPT], an[KC / 4];
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int e = t + q * FH_BLOCK, jj = e / LD, c = e % LD;
            const bool ok = j0 + jj < kd && K0 + j0 + jj < g.n;
            xn[q] = ok ? zs[(size_t)(K0 + j0 + jj) * LD + c] : LU_MK(0, 0);
        }
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            const int jk = j0 + 4 * s + lk, col = K0 + jk;
            an[s] = (irow < r1 && jk < kd && col < g.n) ? A[(size_t)col * N + irow] : LU_MK(0, 0);
        }
    };
    load_chunk(0);
    for (int j0 = 0; j0 < kd; j0 += KC) {
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int e = t + q * FH_BLOCK, jj = e / LD, c = e % LD;
            Xre[jj][c] = xn[q].x; Xim[jj][c] = xn[q].y;
        }
        T ac[KC / 4];
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) ac[s] = an[s];
        __syncthreads();
        if (j0 + KC < kd) load_chunk(j0 + KC);
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
#pragma unroll
            for (int q = 0; q < CT; ++q) {
                if (q >= cta) break;
                const ET xr = Xre[4 * s + lk][16 * q + lr], xi = Xim[4 * s + lk][16 * q + lr];
                re[q] = lu_el<T>::mfma(ac[s].x, xr, re[q]);
                im[q] = lu_el<T>::mfma(ac[s].x, xi, im[q]);
                re[q] = lu_el<T>::mfma(-ac[s].y, xi, re[q]);
                im[q] = lu_el<T>::mfma(ac[s].y, xr, im[q]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < CT; ++q) {
        if (q >= cta) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ib + lu_el<T>::mrow(lk, r);
            if (i < r1) {
                T* d = in + (size_t)i * LD + 16 * q + lr;
                const T y = *d;
                *d = LU_MK(y.x - re[q][r], y.y - im[q][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------
// The side stream of the LU look-ahead (dense and band factorisations): created with a CU mask that leaves `reserve` CUs per
// XCD to the main stream.  Returns false (and switches the look-ahead off for the handle) when no second stream can be had.
// CUs per XCD kept out of the side stream's mask.  One per panel workgroup an XCD receives ((nf + 7) / 8) was the first choice;
// measured later with FH_LU_RESERVE = 2 / 4 / 8: band LU of cfg 3 (16 nodes) 583 / 502 / 500 ms, dense cfg 2 (8 nodes) 79 / 76 /
// 77 ms against 82 at one, cfg 5 (24 nodes) unchanged -- the main stream's small kernels run on the reserved CUs too, and a
// panel workgroup needs a completely empty one.
static int lu_lookahead_reserve(int nf) { (void)nf; return getenv("FH_LU_RESERVE") ? atoi(getenv("FH_LU_RESERVE")) : 4; }
static bool lu_side_stream(feasthip_ctx* h, int reserve) {
    // CUs per XCD left to the main stream: one per panel workgroup the XCD receives (workgroups go round-robin over XCDs)
    if (!h->side_stream || h->side_reserve != reserve) {
        if (h->side_stream) {
            if (hipStreamSynchronize(h->side_stream) != hipSuccess) return false;
            (void)hipStreamDestroy(h->side_stream);
            h->side_stream = nullptr;
        }
        // A panel workgroup (1024 threads x 128 VGPRs) needs a completely empty CU, and the rest update refills every CU
        // as soon as a tile retires, so on a plain (even low-priority) side stream the panels starved until the tail of
        // the update (measured: 676 ms of panel time instead of 123).  The side stream is therefore created with a CU mask
        // that leaves `reserve` CUs per XCD to the main stream.  The reserved set {i : i mod 32 == (i / 32) mod 8 + 8 m,
        // m < reserve} has `reserve` members in every XCD both for an interleaved (XCD = i mod 8, what the kernel driver
        // uses on multi-XCD parts) and a blocked (XCD = i / 32) numbering of the mask bits.
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int i = 0; i < ncu; ++i) {
            bool reserved = false;
            for (int m = 0; m < reserve; ++m) reserved |= (i % 32) == ((i / 32) % 8 + 8 * m);
            if (!reserved) mask[i / 32] |= 1u << (i % 32);
        }
        // a runtime without CU masks (or a partition mode that refuses them) falls back to the plain low-priority stream,
        // and without any second stream to the serial order: the look-ahead is an optimisation, never a reason to fail
        bool have = reserve > 0 && hipExtStreamCreateWithCUMask(&h->side_stream, (uint32_t)mask.size(), mask.data()) == hipSuccess;
        if (!have) {
            (void)hipGetLastError();
            h->side_stream = nullptr;
            int prio_lo = 0, prio_hi = 0;
            have = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess &&
                   hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, prio_lo) == hipSuccess;
        }
        if (have && !h->lu_ev_next)
            have = hipEventCreateWithFlags(&h->lu_ev_next, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&h->lu_ev_rest, hipEventDisableTiming) == hipSuccess;
        if (!have) {
            (void)hipGetLastError();
            if (h->side_stream) { (void)hipStreamDestroy(h->side_stream); h->side_stream = nullptr; }
            h->lu_lookahead = 0;
            return false;
        }
        h->side_reserve = reserve;
    }
    return true;
}

template <typename T>
static int lu_factor_batch(feasthip_ctx* h, const std::vector<int>& which, const std::vector<cplx>& zlist,
                           std::vector<int>& info_out) {
    // which: local node slots to (re)factor; zlist: their shifts
    const int nf = (int)which.size();
    if (nf == 0) return 0;
    const int N = (int)h->dense.N;
    const lu_geom geom = lu_dense_geom<T>(N);
    void* p;
    int rc;
    std::vector<T*> lus(nf);
    std::vector<int*> pvs(nf);
    for (int q = 0; q < nf; ++q) { lus[q] = (T*)h->lu_factors[which[q]]; pvs[q] = h->lu_pivots[which[q]]; }
    if ((rc = fh_get_buf(h, "lu_ptrs", nf * sizeof(T*), &p))) return rc;
    T** dlus = (T**)p;
    if ((rc = fh_get_buf(h, "lu_pptrs", nf * sizeof(int*), &p))) return rc;
    int** dpvs = (int**)p;
    if ((rc = fh_get_buf(h, "lu_z", nf * sizeof(cplx), &p))) return rc;
    cplx* dz = (cplx*)p;
    if ((rc = fh_get_buf(h, "lu_info", nf * sizeof(int), &p))) return rc;
    int* dinfo = (int*)p;
    FH_CHECK(hipMemcpyAsync(dlus, lus.data(), nf * sizeof(T*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(dpvs, pvs.data(), nf * sizeof(int*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(dz, zlist.data(), nf * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemsetAsync(dinfo, 0, nf * sizeof(int), h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));

    // form z B - A
    {
        dim3 grid(2048, nf), block(FH_BLOCK);
        fh_prof_begin(h, "lu_form");
        const bool bid = h->dense.b_identity != 0;
        if (h->dense.is_complex) {
            if (bid) hipLaunchKernelGGL((k_form_shifted<cplx, true, T>), grid, block, 0, h->stream, (const cplx*)h->dense.A, (const cplx*)nullptr, dlus, dz, N);
            else hipLaunchKernelGGL((k_form_shifted<cplx, false, T>), grid, block, 0, h->stream, (const cplx*)h->dense.A, (const cplx*)h->dense.B, dlus, dz, N);
        } else {
            if (bid) hipLaunchKernelGGL((k_form_shifted<double, true, T>), grid, block, 0, h->stream, (const double*)h->dense.A, (const double*)nullptr, dlus, dz, N);
            else hipLaunchKernelGGL((k_form_shifted<double, false, T>), grid, block, 0, h->stream, (const double*)h->dense.A, (const double*)h->dense.B, dlus, dz, N);
        }
        fh_prof_end(h);
    }
    // two-level right-looking LU: outer block columns of LU_KB = 4 LU_NB; inside one, the
    // LU_NB-wide panels update only the block column (k = 32 products on N x <=96), and the
    // trailing matrix gets one k = LU_KB product per block column
    auto laswp = [&](int p0, int np, int a0, int a1, int b0, int b1) {
        const int ncols = (a1 - a0) + (b1 - b0);
        if (ncols <= 0) return;
        fh_prof_begin(h, "lu_laswp");
        hipLaunchKernelGGL((k_lu_laswp<T>), dim3((ncols + FH_BLOCK - 1) / FH_BLOCK, nf), dim3(FH_BLOCK), 0, h->stream, dlus, dpvs, N, p0, np, a0, a1, b0, b1);
        fh_prof_end(h);
    };
    // panels factorised by k_lu_panel_reg leave L11^-1 behind the factor (full 32-column panels only)
    const bool trsm_subst = getenv("FH_LU_TRSM_SUBST") != nullptr;     // comparison: in-place substitution
    auto panel_in_registers = [&](int k0) { return !h->lu_panel_legacy && N - k0 <= 16 * LU_PANEL_THREADS; };
    auto trsm = [&](int k0, int c0, int c1) {
        if (c1 <= c0) return;
        fh_prof_begin(h, "lu_trsm");
        if (!trsm_subst && panel_in_registers(k0) && k0 + LU_NB <= N)
            hipLaunchKernelGGL((k_lu_trsm_mul<LU_NB, T>), dim3((c1 - c0 + FH_BLOCK / LU_NB - 1) / (FH_BLOCK / LU_NB), nf), dim3(FH_BLOCK), 0, h->stream, dlus, geom, k0, c0, c1);
        else
            hipLaunchKernelGGL((k_lu_trsm<LU_NB, T>), dim3((c1 - c0 + FH_BLOCK - 1) / FH_BLOCK, nf), dim3(FH_BLOCK), 0, h->stream, dlus, N, k0, c0, c1);
        fh_prof_end(h);
    };
    auto gemm = [&](int k0, int kd, int r0, int r1, int c0, int c1, const char* cls) {
        if (r1 <= r0 || c1 <= c0) return;
        fh_prof_begin(h, cls);
        if (h->profiling) h->prof_work[cls] += 8.0 * (double)(r1 - r0) * (double)(c1 - c0) * (double)kd * (double)nf;   // real flops of the complex product
        const int TR = (r1 - r0 + 63) / 64, TC = (c1 - c0 + 63) / 64;
        const int sw = std::min(8, TC);
        const int nsuper = ((TR + 7) / 8) * ((TC + sw - 1) / sw);
        // measured on cfg 5: fp64 775 (staged) -> 756 ms (direct); complex64 432 (staged) vs 444 ms (direct)
        const bool staged = h->lu_gemm_staged != 0;
        const dim3 grid(((nsuper + 7) / 8) * 8 * 8 * sw, nf);
        if (staged || sizeof(T) != sizeof(cplx)) hipLaunchKernelGGL((k_lu_gemm<LU_NB, T>), grid, dim3(FH_BLOCK), 0, h->stream, dlus, N, k0, kd, r0, r1, c0, c1, TR, TC);
        else {
            static const bool m3_off = getenv("FH_LU_3M") && atoi(getenv("FH_LU_3M")) == 0;
            if (m3_off) hipLaunchKernelGGL((k_lu_gemm_direct<LU_NB, T, false>), grid, dim3(FH_BLOCK), 0, h->stream, dlus, N, k0, kd, r0, r1, c0, c1, TR, TC, 0);
            else hipLaunchKernelGGL((k_lu_gemm_direct<LU_NB, T, true>), grid, dim3(FH_BLOCK), 0, h->stream, dlus, N, k0, kd, r0, r1, c0, c1, TR, TC, 0);
        }
        fh_prof_end(h);
    };
    // measured (cfg 2 / cfg 5): at N = 4096 a 256-wide outer block is neutral, at N = 8192 it saves 4-6 % (the k = 256
    // trailing product runs at 52 instead of 46 TFLOP/s and outweighs the longer k = 32 in-block updates)
    const int KB = h->lu_outer_block > 0 ? h->lu_outer_block : (N >= 6144 ? 256 : 128);
    // (FH_LU_BLOCKINV=0: the U block row by 32-row products; needs whole 128-row slabs and the register-resident panels' inverses)
    static const bool blockinv_off = getenv("FH_LU_BLOCKINV") && atoi(getenv("FH_LU_BLOCKINV")) == 0;
    const bool block_inverse = !blockinv_off && !trsm_subst && KB % SOLVE_KB == 0 && !h->lu_panel_legacy && N <= 16 * LU_PANEL_THREADS;
    // Look-ahead: the panels of a block column run one 1024-thread workgroup per matrix (8 or 24 of 256 CUs), so the
    // trailing update of block column b is split by columns -- the NEXT block column [Kend, Kend2) is updated on the
    // main stream, the REST [Kend2, N) on a side stream -- and the panels / in-block products of block column b+1
    // overlap the rest.  Hazards: the rest reads the L columns [K0, Kend) and the pivots of block b and writes only
    // columns >= Kend2; block column b+1 writes only [Kend, Kend2) and its own pivots; its interchanges on the other
    // columns (which touch both) wait for the rest.  Same operations on every element, so the factors are bit-identical.
    bool lookahead = h->lu_lookahead != 0 && N > 2 * KB;
    const hipStream_t main_s = h->stream;
    // CUs per XCD left to the main stream: one per panel workgroup the XCD receives (workgroups go round-robin over XCDs)
    const int reserve = lu_lookahead_reserve(nf);
    if (lookahead && !lu_side_stream(h, reserve)) lookahead = false;
    // measured (cfg 2 sweeps): no look-ahead 84 ms, plain side stream 79, plain + 4 chunks 74, CU mask 68, mask + chunks 72
    const int lu_chunks = getenv("FH_LU_CHUNKS") ? std::max(1, atoi(getenv("FH_LU_CHUNKS"))) : (reserve > 0 ? 1 : KB / LU_NB);
    bool rest_pending = false;
    for (int K0 = 0; K0 < N; K0 += KB) {
        const int Kend = std::min(N, K0 + KB);
        for (int k0 = K0; k0 < Kend; k0 += LU_NB) {
            const int nb = std::min(LU_NB, Kend - k0);
            fh_prof_begin(h, "lu_panel");
            {
                const int nrows = N - k0;
                const dim3 g(nf), b(LU_PANEL_THREADS);
                if (!panel_in_registers(k0)) hipLaunchKernelGGL((k_lu_panel<T>), g, b, 0, h->stream, dlus, dpvs, N, k0, nb, dinfo);
                else if (nrows > 8 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<16, 1, T>), g, b, 0, h->stream, dlus, dpvs, geom, N, k0, nb, dinfo, N);
                else if (nrows <= LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<1, 16, T>), g, b, 0, h->stream, dlus, dpvs, geom, N, k0, nb, dinfo, N);
                else if (nrows <= 2 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<2, 8, T>), g, b, 0, h->stream, dlus, dpvs, geom, N, k0, nb, dinfo, N);
                else if (nrows <= 4 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<4, 4, T>), g, b, 0, h->stream, dlus, dpvs, geom, N, k0, nb, dinfo, N);
                else hipLaunchKernelGGL((k_lu_panel_reg<8, 2, T>), g, b, 0, h->stream, dlus, dpvs, geom, N, k0, nb, dinfo, N);
            }
            fh_prof_end(h);
            // interchanges inside the block column (left: finished L columns, right: still to eliminate)
            laswp(k0, nb, K0, k0, k0 + nb, Kend);
            if (k0 + nb < Kend) {     // nb == LU_NB here
                trsm(k0, k0 + nb, Kend);
                gemm(k0, LU_NB, k0 + nb, N, k0 + nb, Kend, "lu_gemm_in");
            }
        }
        if (rest_pending) {           // the previous block's rest update: read the L columns the next call interchanges
            FH_CHECK(hipStreamWaitEvent(main_s, h->lu_ev_rest, 0));
            rest_pending = false;
        }
        // the block column's interchanges, U block row and trailing update on columns [a, b) right of it
        auto right_of_block = [&](int a, int b, int chunks) {
            if (Kend >= N || a >= b) return;   // Kend - K0 == KB from here
            if (block_inverse) {
                // U block row by 128-row slabs: one product with the slab's 128 x 128 inverse of L11 (k_lu_trsm128), one
                // k = 128 update of the slabs below it -- instead of four 32-row products and four k = 32 updates per slab
                for (int k1 = K0; k1 < Kend; k1 += SOLVE_KB) {
                    fh_prof_begin(h, "lu_trsm");
                    hipLaunchKernelGGL((k_lu_trsm128<T>), dim3((b - a + 15) / 16, nf), dim3(FH_BLOCK), 0, h->stream, dlus, geom, k1, a, b);
                    fh_prof_end(h);
                    if (k1 + SOLVE_KB < Kend) gemm(k1, SOLVE_KB, k1 + SOLVE_KB, Kend, a, b, "lu_gemm_in");
                }
            } else
            for (int k0 = K0; k0 < Kend; k0 += LU_NB) {
                trsm(k0, a, b);
                gemm(k0, LU_NB, k0 + LU_NB, Kend, a, b, "lu_gemm_in");
            }
            // in column chunks when it runs beside the next block column: a panel workgroup needs an empty CU and gets
            // one when a chunk drains (an in-order stream starts the next chunk only after the last tile of this one)
            const int step = std::max(512, (((b - a + chunks - 1) / chunks + 511) / 512) * 512);   // whole 8 x 64 super-tile columns
            for (int c = a; c < b; c += step) gemm(K0, KB, Kend, N, c, std::min(b, c + step), "lu_gemm");
        };
        const int Kend2 = lookahead ? std::min(N, Kend + KB) : N;
        if (block_inverse && Kend < N)      // 128 x 128 inverses of the block column's unit-lower diagonal slabs (from the panels' 32-block inverses)
            hipLaunchKernelGGL((k_solve_diag<16, false, true, T>), dim3(SOLVE_KB / 16, (Kend - K0) / SOLVE_KB, nf), dim3(FH_BLOCK), 0, h->stream, dlus,
                               (T*)nullptr, (T*)nullptr, (size_t)0, geom, K0, 0);
        laswp(K0, Kend - K0, 0, K0, Kend, Kend2);
        right_of_block(Kend, Kend2, 1);
        if (Kend2 < N) {
            FH_CHECK(hipEventRecord(h->lu_ev_next, main_s));
            FH_CHECK(hipStreamWaitEvent(h->side_stream, h->lu_ev_next, 0));
            h->stream = h->side_stream;       // the launch helpers and the profiler follow h->stream
            laswp(K0, Kend - K0, 0, 0, Kend2, N);
            right_of_block(Kend2, N, lu_chunks);
            const hipError_t er = hipEventRecord(h->lu_ev_rest, h->side_stream);
            h->stream = main_s;
            FH_CHECK(er);
            rest_pending = true;
        }
    }
    if (rest_pending) FH_CHECK(hipStreamWaitEvent(main_s, h->lu_ev_rest, 0));
    fh_prof_begin(h, "lu_invert");
    hipLaunchKernelGGL((k_lu_invert_diag<LU_NB, T>), dim3((N + LU_NB - 1) / LU_NB, nf), dim3(64), 0, h->stream, dlus, geom);
    {   // 128 x 128 inverses from the 32-block inverses: the in-block substitution applied to the identity
        dim3 g(SOLVE_KB / 16, (N + SOLVE_KB - 1) / SOLVE_KB, nf);
        hipLaunchKernelGGL((k_solve_diag<16, false, true, T>), g, dim3(FH_BLOCK), 0, h->stream, dlus, (T*)nullptr, (T*)nullptr, (size_t)0, geom, 0, 0);
        hipLaunchKernelGGL((k_solve_diag<16, true, true, T>), g, dim3(FH_BLOCK), 0, h->stream, dlus, (T*)nullptr, (T*)nullptr, (size_t)0, geom, 0, 0);
    }
    if (N <= 16000) hipLaunchKernelGGL(k_build_perm, dim3(nf), dim3(FH_BLOCK), (size_t)N * sizeof(int), h->stream, dpvs, N);
    else hipLaunchKernelGGL(k_build_perm_global, dim3(nf), dim3(64), 0, h->stream, dpvs, N);
    fh_prof_end(h);
    info_out.assign(nf, 0);
    FH_CHECK(hipMemcpyAsync(info_out.data(), dinfo, nf * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}

// rows per workgroup (in units of 64): enough workgroups to fill the chip, few enough that the
// per-workgroup slab product is amortised
static int lu_solve_rb(int rows, int nf) {
    int rb = (int)(((long long)rows * nf) / (64LL * 768));
    return std::max(1, std::min(8, rb));
}

template <int LD, typename T>
static void lu_solve_launch(feasthip_ctx* h, T** dlus, T* Y, T* Z, size_t stride, int N, int nf, int m) {
    const int cta = std::max(1, std::min(LD / 16, (m + 15) / 16));
    const lu_geom geom = lu_dense_geom<T>(N);
    const int nblocks = (N + LU_NB - 1) / LU_NB;
    const bool one_level = h->lu_solve_legacy != 0;
    if (!one_level) {
        const int nouter = (N + SOLVE_KB - 1) / SOLVE_KB;
        for (int b = 0; b < nouter; ++b) {        // forward: L z = P b   (Y -> Z)
            const int K0 = b * SOLVE_KB, kb = std::min(SOLVE_KB / LU_NB, (N - K0 + LU_NB - 1) / LU_NB);
            const int r0 = K0 + LU_NB * kb;
            hipLaunchKernelGGL((k_solve_diag_inv<LD, false, T>), dim3(cta, nf), dim3(FH_BLOCK), 0, h->stream, dlus, Y, Z, stride, geom, K0, kb);
            if (r0 < N)
                hipLaunchKernelGGL((k_solve_update<LD, T>), dim3((N - r0 + 63) / 64, nf), dim3(FH_BLOCK), 0, h->stream, dlus, Y, Z,
                                   stride, geom, K0, LU_NB * kb, r0, N, cta);
        }
        for (int b = nouter - 1; b >= 0; --b) {   // backward: U x = z   (Z -> Y)
            const int K0 = b * SOLVE_KB, kb = std::min(SOLVE_KB / LU_NB, (N - K0 + LU_NB - 1) / LU_NB);
            hipLaunchKernelGGL((k_solve_diag_inv<LD, true, T>), dim3(cta, nf), dim3(FH_BLOCK), 0, h->stream, dlus, Z, Y, stride, geom, K0, kb);
            if (K0 > 0)
                hipLaunchKernelGGL((k_solve_update<LD, T>), dim3((K0 + 63) / 64, nf), dim3(FH_BLOCK), 0, h->stream, dlus, Z, Y,
                                   stride, geom, K0, LU_NB * kb, 0, K0, cta);
        }
        return;
    }
    for (int b = 0; b < nblocks; ++b) {        // forward: L z = P b   (Y -> Z)
        const int k0 = b * LU_NB, r0 = std::min(N, k0 + LU_NB);
        const int rb = lu_solve_rb(N - r0, nf);
        const int gx = std::max(1, (N - r0 + 64 * rb - 1) / (64 * rb));
        hipLaunchKernelGGL((k_solve_step<LU_NB, LD, false, T>), dim3(gx, nf), dim3(FH_BLOCK), 0, h->stream, dlus, Y, Z, stride, N, k0, r0, N, rb, cta);
    }
    for (int b = nblocks - 1; b >= 0; --b) {   // backward: U x = z   (Z -> Y)
        const int k0 = b * LU_NB;
        const int rb = lu_solve_rb(k0, nf);
        const int gx = std::max(1, (k0 + 64 * rb - 1) / (64 * rb));
        hipLaunchKernelGGL((k_solve_step<LU_NB, LD, true, T>), dim3(gx, nf), dim3(FH_BLOCK), 0, h->stream, dlus, Z, Y, stride, N, k0, 0, k0, rb, cta);
    }
}

// Solve with the cached factors of `slots`.  RHS: fp64 panel(s), rhs_stride = 0 when one panel is shared by
// all nodes; Y: fp64 output panels.  T = cplxf: the right-hand side is narrowed while it is permuted, the
// substitutions run in complex64 and the result is widened into Y (one step of the refinement loop).
template <typename T>
static int lu_solve_batch(feasthip_ctx* h, int ld, int m, const std::vector<int>& slots, const cplx* RHS, size_t rhs_stride,
                          cplx* Y, size_t stride) {
    const int nf = (int)slots.size();
    const int N = (int)h->dense.N;
    void* p;
    int rc;
    std::vector<T*> lus(nf);
    std::vector<int*> perms(nf);
    for (int q = 0; q < nf; ++q) { lus[q] = (T*)h->lu_factors[slots[q]]; perms[q] = h->lu_pivots[slots[q]] + N; }
    if ((rc = fh_get_buf(h, "lu_ptrs", nf * sizeof(T*), &p))) return rc;
    T** dlus = (T**)p;
    if ((rc = fh_get_buf(h, "lu_permptrs", nf * sizeof(int*), &p))) return rc;
    int** dperms = (int**)p;
    if ((rc = fh_get_buf(h, "lu_zpanel", (size_t)nf * stride * sizeof(T), &p))) return rc;
    T* Z = (T*)p;
    T* W = (T*)Y;                                   // working panel: Y itself in fp64, a complex64 buffer otherwise
    if (sizeof(T) != sizeof(cplx)) {
        if ((rc = fh_get_buf(h, "lu_wpanel", (size_t)nf * stride * sizeof(T), &p))) return rc;
        W = (T*)p;
    }
    FH_CHECK(hipMemcpyAsync(dlus, lus.data(), nf * sizeof(T*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(dperms, perms.data(), nf * sizeof(int*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    fh_prof_begin(h, "lu_solve");
    hipLaunchKernelGGL((k_gather_rows<T>), dim3(fh_vec_nblk(N, ld), nf), dim3(FH_BLOCK), 0, h->stream, RHS, rhs_stride, dperms, W, stride, N, ld);
    if (ld == 16) lu_solve_launch<16, T>(h, dlus, W, Z, stride, N, nf, m);
    else if (ld == 32) lu_solve_launch<32, T>(h, dlus, W, Z, stride, N, nf, m);
    else lu_solve_launch<64, T>(h, dlus, W, Z, stride, N, nf, m);
    if constexpr (sizeof(T) != sizeof(cplx))
        hipLaunchKernelGGL(k_widen_panels, dim3(fh_vec_nblk(N, ld), nf), dim3(FH_BLOCK), 0, h->stream, (const cplxf*)W, Y, stride, (size_t)N * ld);
    fh_prof_end(h);
    return 0;
}

static int lu_ensure_slots(feasthip_ctx* h, int nslots) {
    const size_t N = (size_t)h->dense.N;
    const int prec = h->factor_precision == 32 ? 32 : 64;
    if (h->lu_prec != prec) {                       // factor precision changed: drop the cached factors
        for (void* p : h->lu_factors) if (p) hipFree(p);
        for (int* p : h->lu_pivots) if (p) hipFree(p);
        h->lu_factors.clear(); h->lu_pivots.clear(); h->lu_valid.clear(); h->lu_z.clear();
        h->lu_prec = prec;
    }
    const size_t esz = prec == 32 ? sizeof(cplxf) : sizeof(cplx);
    while ((int)h->lu_factors.size() < nslots) {
        void* f = nullptr; int* pv = nullptr;
        // factor, then the inverted diagonal blocks (k_lu_invert_diag); pivots, then the row permutation
        const size_t nblk = (N + LU_NB - 1) / LU_NB;
        const size_t nblk128 = (N + SOLVE_KB - 1) / SOLVE_KB;
        FH_CHECK(hipMalloc(&f, (N * N + nblk * 2 * LU_NB * LU_NB + nblk128 * 2 * SOLVE_KB * SOLVE_KB) * esz));
        hipError_t e = hipMalloc((void**)&pv, 2 * N * sizeof(int));
        if (e != hipSuccess) { hipFree(f); h->last_error = "hipMalloc(pivots)"; return FEASTHIP_ERROR_MEMORY; }
        h->lu_factors.push_back(f); h->lu_pivots.push_back(pv); h->lu_valid.push_back(0); h->lu_z.push_back(cmake(0, 0));
    }
    return 0;
}

int fh_dense_lu_solve_nodes(feasthip_ctx* h, int ld, int m, int nodes, const std::vector<cplx>& z, const cplx* RHS,
                            size_t rhs_stride, cplx* Y, size_t stride, std::vector<int>& status, int64_t* nfact) {
    int rc = lu_ensure_slots(h, nodes);
    const bool f32 = h->lu_prec == 32;
    if (rc) return rc;
    std::vector<int> need;
    std::vector<cplx> zl;
    for (int e = 0; e < nodes; ++e) {
        bool ok = h->cache_factors && h->lu_valid[e] == 1 && h->lu_z[e].x == z[e].x && h->lu_z[e].y == z[e].y;
        if (!ok) { need.push_back(e); zl.push_back(z[e]); h->lu_valid[e] = 0; }
    }
    std::vector<int> info;
    if ((rc = f32 ? lu_factor_batch<cplxf>(h, need, zl, info) : lu_factor_batch<cplx>(h, need, zl, info))) return rc;
    for (size_t q = 0; q < need.size(); ++q) {
        h->lu_z[need[q]] = zl[q];
        h->lu_valid[need[q]] = info[q] == 0 ? 1 : -1;   // -1: singular
    }
    if (nfact) *nfact = (int64_t)need.size();
    std::vector<int> slots(nodes);
    for (int e = 0; e < nodes; ++e) slots[e] = e;
    if ((rc = f32 ? lu_solve_batch<cplxf>(h, ld, m, slots, RHS, rhs_stride, Y, stride)
                  : lu_solve_batch<cplx>(h, ld, m, slots, RHS, rhs_stride, Y, stride))) return rc;
    status.assign(nodes, 0);
    for (int e = 0; e < nodes; ++e) if (h->lu_valid[e] != 1) status[e] = FEASTHIP_ERROR_LAPACK;
    return 0;
}

int fh_dense_lu_solve_single(feasthip_ctx* h, int ld, int m, cplx z, const cplx* RHS, cplx* Y, int* status, int64_t* nfact) {
    // Factor cache for one-off shifts (RCI jobs 10/11, linear_solver callbacks): a shift equal to
    // a local quadrature node uses that node's slot, so an RCI sweep over the contour keeps one
    // factorisation per node across refinement loops, as the reference's drivers do
    // (factor_cache, src/dense/feast_dense.jl:458, 487-497); any other shift shares one extra slot.
    int slot = h->node_count;
    for (int e = 0; e < h->node_count && e < (int)h->node_ids.size(); ++e) {
        const cplx ze = h->zne[h->node_ids[e]];
        if (ze.x == z.x && ze.y == z.y) { slot = e; break; }
    }
    int rc = lu_ensure_slots(h, std::max(slot, h->node_count) + 1);
    if (rc) return rc;
    const bool f32 = h->lu_prec == 32;
    std::vector<int> need(1, slot), info;
    std::vector<cplx> zl(1, z);
    bool cached = h->cache_factors && h->lu_valid[slot] == 1 && h->lu_z[slot].x == z.x && h->lu_z[slot].y == z.y;
    if (!cached) {
        if ((rc = f32 ? lu_factor_batch<cplxf>(h, need, zl, info) : lu_factor_batch<cplx>(h, need, zl, info))) return rc;
        h->lu_z[slot] = z;
        h->lu_valid[slot] = info[0] == 0 ? 1 : -1;
        if (nfact) *nfact = 1;
    }
    if ((rc = f32 ? lu_solve_batch<cplxf>(h, ld, m, need, RHS, 0, Y, (size_t)h->dense.N * ld)
                  : lu_solve_batch<cplx>(h, ld, m, need, RHS, 0, Y, (size_t)h->dense.N * ld))) return rc;
    *status = h->lu_valid[slot] == 1 ? 0 : FEASTHIP_ERROR_LAPACK;
    return 0;
}

// =======================================================================================
// Blocked band LU on the dense kernels: the sparse DIRECT solver for patterns whose band (after the reverse
// Cuthill-McKee renumbering of fh_ingest.hpp) is too wide for the one-workgroup-per-node elimination of fh_banded.hip.
// Reference: the sparse drivers' default `lu(z B - A)` (UMFPACK, src/sparse/feast_sparse.jl:339, 342; parallel workers
// src/parallel/feast_parallel.jl:603-613, 728-735) -- a direct factorisation per quadrature node, cached across refinement
// loops.  UMFPACK's multifrontal elimination is replaced by a band elimination (same arithmetic class: Gaussian
// elimination with partial pivoting, fill confined to the band instead of the elimination tree).
//
// Storage per node: general band storage with room for the blocks,
//     AB(ldab, N),  ldab = 2 kl + ku + 2 WB,  A(i, j) = AB[kvp + i - j + j ldab],  kvp = kl + ku + WB,  WB = 128,
// i.e. LAPACK's ZGBTRF layout with WB extra rows above and below.  Seen through the pointer base = AB + kvp and the leading
// dimension ldab - 1 this is an ordinary column-major matrix, A(i, j) = base[i + j (ldab - 1)], in which every entry of the
// window of block column [K0, K0 + WB) -- rows up to K0 + WB + kl, columns up to K0 + WB + kl + ku -- has its own storage
// cell (entries outside band + fill are zeros that stay zero).  The elimination is therefore the dense two-level LU of this
// file restricted to that window: the register-resident panels, the MFMA trailing update, the inverted diagonal blocks, all
// unchanged, told the geometry through lu_geom.  As in ZGBTRF the interchanges of a block column are NOT applied to the
// block columns left of it; the substitution applies them block by block (k_wband_swap).
// Work: 8 N kl (kl + ku) flop per node (cfg 3 after RCM: kl = ku = 951, 7.2e11), storage 16 N ldab bytes (2.5 GB).
// =======================================================================================
#define WB 128
static_assert(WB == SOLVE_KB && WB % LU_NB == 0, "factor blocks = substitution blocks");

struct wband_geom { int N, kl, ku, ldab, kvp; lu_geom g; size_t elems; };
static wband_geom wband_geometry(int N, int kl, int ku) {
    wband_geom w;
    w.N = N; w.kl = kl; w.ku = ku;
    w.ldab = 2 * kl + ku + 2 * WB;
    w.kvp = kl + ku + WB;
    const size_t nblk32 = ((size_t)N + LU_NB - 1) / LU_NB, nblk128 = ((size_t)N + SOLVE_KB - 1) / SOLVE_KB;
    w.g.ld = w.ldab - 1;
    w.g.n = N;
    w.g.inv32 = (size_t)w.ldab * N - w.kvp;
    w.g.inv128 = w.g.inv32 + nblk32 * 2 * LU_NB * LU_NB;
    w.elems = (size_t)w.ldab * N + nblk32 * 2 * LU_NB * LU_NB + nblk128 * 2 * SOLVE_KB * SOLVE_KB;
    return w;
}
size_t fh_wband_elems(int N, int kl, int ku) { return wband_geometry(N, kl, ku).elems; }
size_t fh_wband_base_offset(int N, int kl, int ku) { return (size_t)wband_geometry(N, kl, ku).kvp; }

// base[i' + j' ld] = z B - A for the renumbered unknowns i' = iperm[i] (iperm null: as stored); the storage was zeroed
template <typename VT, bool BIDENT, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_wband_form(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          const VT* __restrict__ aval, const VT* __restrict__ bval,
                                                          T* const* bases, const cplx* z, const int* __restrict__ iperm, int N, int ld) {
    T* base = bases[blockIdx.y];
    const cplx zz = z[blockIdx.y];
    const int i = blockIdx.x * FH_BLOCK + threadIdx.x;
    if (i >= N) return;
    const int bi = iperm ? iperm[i] : i;
    if (BIDENT) base[(size_t)bi * ld + bi] = cvt<T>(zz);
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int bj = iperm ? iperm[col[k]] : col[k];
        cplx a;
        if constexpr (sizeof(VT) == sizeof(cplx)) a = cmake(aval[k].x, aval[k].y); else a = cmake(aval[k], 0.0);
        T* dst = base + (size_t)bj * ld + bi;
        if (BIDENT) {
            *dst = cvt<T>(csub(to_d(*dst), a));
        } else {
            cplx b;
            if constexpr (sizeof(VT) == sizeof(cplx)) b = cmake(bval[k].x, bval[k].y); else b = cmake(bval[k], 0.0);
            *dst = cvt<T>(csub(cmul(zz, b), a));
        }
    }
}

// The row interchanges piv[K0 .. K0 + nbk) of one block column applied to the rows [K0, K0 + W) of row-major panels:
// the swaps are composed on row INDICES in LDS (one thread, nbk short steps), the (at most 2 nbk) rows that end up
// somewhere else are read by all threads into registers and written to their places.  grid (ld / 16, nodes).
template <typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_wband_swap(int* const* pivs, T* Y, size_t stride, int ld, int K0, int nbk, int W) {
    extern __shared__ int wb_cur[];          // [W] original row (relative to K0) now at position i
    __shared__ int s_dst[2 * WB], s_src[2 * WB];
    __shared__ int s_cnt;
    const int* piv = pivs[blockIdx.y] + K0;
    T* Yn = Y + (size_t)blockIdx.y * stride + 16 * blockIdx.x;
    const int t = threadIdx.x;
    for (int i = t; i < W; i += FH_BLOCK) wb_cur[i] = i;
    if (t == 0) s_cnt = 0;
    __syncthreads();
    if (t == 0) {
        for (int j = 0; j < nbk; ++j) {
            const int p = piv[j] - K0;
            if (p != j && p >= 0 && p < W) { const int u = wb_cur[j]; wb_cur[j] = wb_cur[p]; wb_cur[p] = u; }
        }
    }
    __syncthreads();
    for (int i = t; i < W; i += FH_BLOCK) {
        if (wb_cur[i] != i) { const int q = atomicAdd(&s_cnt, 1); if (q < 2 * WB) { s_dst[q] = i; s_src[q] = wb_cur[i]; } }
    }
    __syncthreads();
    const int cnt = min(s_cnt, 2 * WB);
    const int c = t & 15, rr = t >> 4;
    constexpr int RL = FH_BLOCK / 16, NV = 2 * WB / RL;
    T v[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int e = rr + q * RL;
        v[q] = e < cnt ? Yn[(size_t)(K0 + s_src[e]) * ld + c] : LU_MK(0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int e = rr + q * RL;
        if (e < cnt) Yn[(size_t)(K0 + s_dst[e]) * ld + c] = v[q];
    }
}

// Y[node][perm[i], :] = Yb[node][i, :]
template <typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_scatter_rows(const T* __restrict__ Yb, size_t bstride, const int* __restrict__ perm,
                                                            cplx* __restrict__ Y, size_t stride, int N, int ld) {
    const T* s = Yb + (size_t)blockIdx.y * bstride;
    cplx* d = Y + (size_t)blockIdx.y * stride;
    const size_t total = (size_t)N * ld;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const size_t i = e / ld, c = e % ld;
        d[(size_t)(perm ? perm[i] : (int)i) * ld + c] = to_d(s[e]);
    }
}

// Form z B - A in band storage (zeroed here) and factor it, nf nodes at once.  dabs: device array of the nf storage pointers
// (AB, not base), dbases: the same plus kvp; dpvs: pivots (N ints per node, global row indices); info_out[q] = 0 or the
// 1-based column of a zero pivot.
template <typename T>
static int wband_factor_t(feasthip_ctx* h, int nf, void* const* abs_host, T** dbases, int** dpvs, const cplx* dz, int* dinfo,
                          const int* d_iperm, int kl, int ku) {
    const int N = (int)h->csr.N;
    const wband_geom w = wband_geometry(N, kl, ku);
    const lu_geom geom = w.g;
    const int lda = geom.ld;
    fh_prof_begin(h, "wband_form");
    for (int q = 0; q < nf; ++q) FH_CHECK(hipMemsetAsync(abs_host[q], 0, (size_t)w.ldab * N * sizeof(T), h->stream));
    {
        const dim3 grid((N + FH_BLOCK - 1) / FH_BLOCK, nf), block(FH_BLOCK);
        const bool bid = h->csr.b_identity != 0;
        if (h->csr.is_complex) {
            if (bid) hipLaunchKernelGGL((k_wband_form<cplx, true, T>), grid, block, 0, h->stream, h->csr.rowptr, h->csr.col, (const cplx*)h->csr.aval, (const cplx*)nullptr, dbases, dz, d_iperm, N, lda);
            else hipLaunchKernelGGL((k_wband_form<cplx, false, T>), grid, block, 0, h->stream, h->csr.rowptr, h->csr.col, (const cplx*)h->csr.aval, (const cplx*)h->csr.bval, dbases, dz, d_iperm, N, lda);
        } else {
            if (bid) hipLaunchKernelGGL((k_wband_form<double, true, T>), grid, block, 0, h->stream, h->csr.rowptr, h->csr.col, (const double*)h->csr.aval, (const double*)nullptr, dbases, dz, d_iperm, N, lda);
            else hipLaunchKernelGGL((k_wband_form<double, false, T>), grid, block, 0, h->stream, h->csr.rowptr, h->csr.col, (const double*)h->csr.aval, (const double*)h->csr.bval, dbases, dz, d_iperm, N, lda);
        }
    }
    fh_prof_end(h);
    fh_prof_begin(h, "wband_lu");
    auto laswp = [&](int p0, int np, int a0, int a1, int b0, int b1) {
        const int ncols = (a1 - a0) + (b1 - b0);
        if (ncols <= 0) return;
        hipLaunchKernelGGL((k_lu_laswp<T>), dim3((ncols + FH_BLOCK - 1) / FH_BLOCK, nf), dim3(FH_BLOCK), 0, h->stream, dbases, dpvs, lda, p0, np, a0, a1, b0, b1);
    };
    auto trsm = [&](int k0, int nb, int c0, int c1) {
        if (c1 <= c0) return;
        if (nb == LU_NB)
            hipLaunchKernelGGL((k_lu_trsm_mul<LU_NB, T>), dim3((c1 - c0 + FH_BLOCK / LU_NB - 1) / (FH_BLOCK / LU_NB), nf), dim3(FH_BLOCK), 0, h->stream, dbases, geom, k0, c0, c1);
        else
            hipLaunchKernelGGL((k_lu_trsm<LU_NB, T>), dim3((c1 - c0 + FH_BLOCK - 1) / FH_BLOCK, nf), dim3(FH_BLOCK), 0, h->stream, dbases, lda, k0, c0, c1);
    };
    static const bool m3_off = getenv("FH_LU_3M") && atoi(getenv("FH_LU_3M")) == 0;
    auto gemm = [&](int k0, int kd, int r0, int r1, int c0, int c1) {
        if (r1 <= r0 || c1 <= c0) return;
        if (h->profiling) h->prof_work["wband_lu"] += 8.0 * (double)(r1 - r0) * (double)(c1 - c0) * (double)kd * (double)nf;
        const int TR = (r1 - r0 + 63) / 64, TC = (c1 - c0 + 63) / 64;
        const int sw = std::min(8, TC);
        const int nsuper = ((TR + 7) / 8) * ((TC + sw - 1) / sw);
        const dim3 grid(((nsuper + 7) / 8) * 8 * 8 * sw, nf);
        if constexpr (sizeof(T) != sizeof(cplx)) hipLaunchKernelGGL((k_lu_gemm<LU_NB, T>), grid, dim3(FH_BLOCK), 0, h->stream, dbases, lda, k0, kd, r0, r1, c0, c1, TR, TC);
        else if (m3_off) hipLaunchKernelGGL((k_lu_gemm_direct<LU_NB, T, false>), grid, dim3(FH_BLOCK), 0, h->stream, dbases, lda, k0, kd, r0, r1, c0, c1, TR, TC, 0);
        else hipLaunchKernelGGL((k_lu_gemm_direct<LU_NB, T, true>), grid, dim3(FH_BLOCK), 0, h->stream, dbases, lda, k0, kd, r0, r1, c0, c1, TR, TC, 0);
    };
    // Look-ahead, as in the dense factorisation: the panels of a block column run one workgroup per node, so the update right
    // of block column b is split by columns -- the NEXT block column [Kend, Kend + WB) on the main stream, the REST
    // [Kend + WB, nc) on the CU-masked side stream -- and the panels of block column b+1 overlap the rest.  The rest reads the
    // L columns and pivots of block b and writes only columns >= Kend + WB; block column b+1 writes only its own columns and
    // pivots until it waits for the rest (its interchanges to the right touch the same columns).  No interchanges go to the
    // left in the band factorisation, so nothing else is shared.  Same operations on every element: identical factors.
    static const bool block_inverse = !(getenv("FH_WBAND_BLOCKINV") && atoi(getenv("FH_WBAND_BLOCKINV")) == 0);
    bool lookahead = h->lu_lookahead != 0 && kl + ku > 2 * WB && N > 4 * WB;
    if (lookahead && !lu_side_stream(h, lu_lookahead_reserve(nf))) lookahead = false;
    const hipStream_t main_s = h->stream;
    bool rest_pending = false;
    for (int K0 = 0; K0 < N; K0 += WB) {
        const int Kend = std::min(N, K0 + WB);
        const int nr = std::min(N, Kend + kl);                 // rows the block column reaches
        const int nc = std::min(N, Kend + kl + ku);            // columns its pivot rows reach
        for (int k0 = K0; k0 < Kend; k0 += LU_NB) {
            const int nb = std::min(LU_NB, Kend - k0);
            const int nrows = nr - k0;
            const dim3 g(nf), b(LU_PANEL_THREADS);
            if (nrows > 8 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<16, 1, T>), g, b, 0, h->stream, dbases, dpvs, geom, nr, k0, nb, dinfo, nr);
            else if (nrows <= LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<1, 16, T>), g, b, 0, h->stream, dbases, dpvs, geom, nr, k0, nb, dinfo, nr);
            else if (nrows <= 2 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<2, 8, T>), g, b, 0, h->stream, dbases, dpvs, geom, nr, k0, nb, dinfo, nr);
            else if (nrows <= 4 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<4, 4, T>), g, b, 0, h->stream, dbases, dpvs, geom, nr, k0, nb, dinfo, nr);
            else hipLaunchKernelGGL((k_lu_panel_reg<8, 2, T>), g, b, 0, h->stream, dbases, dpvs, geom, nr, k0, nb, dinfo, nr);
            laswp(k0, nb, K0, k0, k0 + nb, Kend);              // inside the block column only
            if (k0 + nb < Kend) {
                trsm(k0, nb, k0 + nb, Kend);
                gemm(k0, LU_NB, k0 + nb, nr, k0 + nb, Kend);
            }
        }
        if (rest_pending) {
            if (hipStreamWaitEvent(main_s, h->lu_ev_rest, 0) != hipSuccess) { h->stream = main_s; h->last_error = "hipStreamWaitEvent(band LU)"; return FEASTHIP_ERROR_INTERNAL; }
            rest_pending = false;
        }
        if (Kend >= N) break;
        // interchanges (to the right only: ZGBTRF), U block row and update of the columns [a, b) right of the block column
        auto right_of_block = [&](int a, int b) {
            if (a >= b) return;
            laswp(K0, Kend - K0, 0, 0, a, b);
            if (block_inverse) {                               // Kend - K0 == WB here
                hipLaunchKernelGGL((k_lu_trsm128<T>), dim3((b - a + 15) / 16, nf), dim3(FH_BLOCK), 0, h->stream, dbases, geom, K0, a, b);
            } else {
                for (int k0 = K0; k0 < Kend; k0 += LU_NB) {
                    trsm(k0, LU_NB, a, b);
                    gemm(k0, LU_NB, k0 + LU_NB, Kend, a, b);
                }
            }
            gemm(K0, WB, Kend, nr, a, b);
        };
        // the 128 x 128 inverse of the block column's unit-lower L11 (from the 32-block inverses the panels left): the U block
        // row right of it is then ONE product per side of the look-ahead instead of four 32-row products and three k = 32
        // row-block products (FH_WBAND_BLOCKINV=0: the 32-row sequence)
        if (block_inverse)
            hipLaunchKernelGGL((k_solve_diag<16, false, true, T>), dim3(SOLVE_KB / 16, 1, nf), dim3(FH_BLOCK), 0, h->stream, dbases, (T*)nullptr, (T*)nullptr, (size_t)0, geom, K0, 0);
        const int Kend2 = lookahead ? std::min(nc, Kend + WB) : nc;
        // the rest needs the block column's panels only, not the next block column's update: it starts beside that update
        if (Kend2 < nc) {
            hipError_t er = hipEventRecord(h->lu_ev_next, main_s);
            if (er == hipSuccess) er = hipStreamWaitEvent(h->side_stream, h->lu_ev_next, 0);
            if (er != hipSuccess) { h->last_error = "hipEventRecord(band LU)"; return FEASTHIP_ERROR_INTERNAL; }
            h->stream = h->side_stream;                        // the launch helpers follow h->stream
            right_of_block(Kend2, nc);
            er = hipEventRecord(h->lu_ev_rest, h->side_stream);
            h->stream = main_s;
            if (er != hipSuccess) { h->last_error = "hipEventRecord(band LU)"; return FEASTHIP_ERROR_INTERNAL; }
            rest_pending = true;
        }
        right_of_block(Kend, Kend2);
    }
    if (rest_pending && hipStreamWaitEvent(main_s, h->lu_ev_rest, 0) != hipSuccess) { h->last_error = "hipStreamWaitEvent(band LU)"; return FEASTHIP_ERROR_INTERNAL; }
    hipLaunchKernelGGL((k_lu_invert_diag<LU_NB, T>), dim3((N + LU_NB - 1) / LU_NB, nf), dim3(64), 0, h->stream, dbases, geom);
    {
        dim3 g(SOLVE_KB / 16, (N + SOLVE_KB - 1) / SOLVE_KB, nf);
        hipLaunchKernelGGL((k_solve_diag<16, false, true, T>), g, dim3(FH_BLOCK), 0, h->stream, dbases, (T*)nullptr, (T*)nullptr, (size_t)0, geom, 0, 0);
        hipLaunchKernelGGL((k_solve_diag<16, true, true, T>), g, dim3(FH_BLOCK), 0, h->stream, dbases, (T*)nullptr, (T*)nullptr, (size_t)0, geom, 0, 0);
    }
    fh_prof_end(h);
    (void)dinfo;
    return 0;
}

template <int LD, typename T>
static void wband_solve_launch(feasthip_ctx* h, T** dbases, int** dpvs, T* Y, T* Z, size_t stride, const wband_geom& w, int nf, int m) {
    const int N = w.N, kl = w.kl, kv = w.kl + w.ku;
    const lu_geom geom = w.g;
    const int cta = std::max(1, std::min(LD / 16, (m + 15) / 16));
    const int nouter = (N + SOLVE_KB - 1) / SOLVE_KB;
    for (int b = 0; b < nouter; ++b) {             // forward: interchanges of the block, L11 z = y, rows below -= L21 z   (Y -> Z)
        const int K0 = b * SOLVE_KB, Kend = std::min(N, K0 + SOLVE_KB);
        const int kb = (Kend - K0 + LU_NB - 1) / LU_NB;
        const int nr = std::min(N, Kend + kl);
        hipLaunchKernelGGL((k_wband_swap<T>), dim3(LD / 16, nf), dim3(FH_BLOCK), (size_t)(nr - K0) * sizeof(int), h->stream, dpvs, Y, stride, LD, K0, Kend - K0, nr - K0);
        hipLaunchKernelGGL((k_solve_diag_inv<LD, false, T>), dim3(cta, nf), dim3(FH_BLOCK), 0, h->stream, dbases, Y, Z, stride, geom, K0, kb);
        if (Kend < nr)
            hipLaunchKernelGGL((k_solve_update<LD, T>), dim3((nr - Kend + 63) / 64, nf), dim3(FH_BLOCK), 0, h->stream, dbases, Y, Z, stride, geom, K0, Kend - K0, Kend, nr, cta);
    }
    for (int b = nouter - 1; b >= 0; --b) {        // backward: U11 x = z, rows above (within kl + ku) -= U12 x   (Z -> Y)
        const int K0 = b * SOLVE_KB, Kend = std::min(N, K0 + SOLVE_KB);
        const int kb = (Kend - K0 + LU_NB - 1) / LU_NB;
        hipLaunchKernelGGL((k_solve_diag_inv<LD, true, T>), dim3(cta, nf), dim3(FH_BLOCK), 0, h->stream, dbases, Z, Y, stride, geom, K0, kb);
        const int r0 = std::max(0, K0 - kv);
        if (r0 < K0)
            hipLaunchKernelGGL((k_solve_update<LD, T>), dim3((K0 - r0 + 63) / 64, nf), dim3(FH_BLOCK), 0, h->stream, dbases, Z, Y, stride, geom, K0, Kend - K0, r0, K0, cta);
    }
}

// Y[node] = (z_node B - A)^-1 RHS with the factors of fh_wband_factor.  RHS: one shared panel (row-major N x ld, the
// library's row order); d_perm[band row] = library row (null: same order); Yb, Zb: nf work panels in band order.
template <typename T>
static int wband_solve_t(feasthip_ctx* h, int nf, T** dbases, int** dpvs, int** dperms, const int* d_perm, const cplx* RHS, size_t rhs_stride,
                         cplx* Y, size_t stride, T* Yb, T* Zb, int ld, int m, int kl, int ku) {
    const int N = (int)h->csr.N;
    const wband_geom w = wband_geometry(N, kl, ku);
    const size_t bstride = (size_t)N * ld;
    fh_prof_begin(h, "wband_solve");
    hipLaunchKernelGGL((k_gather_rows<T>), dim3(fh_vec_nblk(N, ld), nf), dim3(FH_BLOCK), 0, h->stream, RHS, rhs_stride, dperms, Yb, bstride, N, ld);
    if (ld == 16) wband_solve_launch<16, T>(h, dbases, dpvs, Yb, Zb, bstride, w, nf, m);
    else if (ld == 32) wband_solve_launch<32, T>(h, dbases, dpvs, Yb, Zb, bstride, w, nf, m);
    else wband_solve_launch<64, T>(h, dbases, dpvs, Yb, Zb, bstride, w, nf, m);
    hipLaunchKernelGGL((k_scatter_rows<T>), dim3(fh_vec_nblk(N, ld), nf), dim3(FH_BLOCK), 0, h->stream, Yb, bstride, d_perm, Y, stride, N, ld);
    fh_prof_end(h);
    return 0;
}

// exported entry points: prec = 64 (complex128 factors) or 32 (complex64 factors; the caller refines in fp64)
int fh_wband_factor(feasthip_ctx* h, int prec, int nf, void* const* abs_host, void** dbases, int** dpvs, const cplx* dz, int* dinfo,
                    const int* d_iperm, int kl, int ku) {
    if (prec == 32) return wband_factor_t<cplxf>(h, nf, abs_host, (cplxf**)dbases, dpvs, dz, dinfo, d_iperm, kl, ku);
    return wband_factor_t<cplx>(h, nf, abs_host, (cplx**)dbases, dpvs, dz, dinfo, d_iperm, kl, ku);
}
int fh_wband_solve(feasthip_ctx* h, int prec, int nf, void** dbases, int** dpvs, int** dperms, const int* d_perm, const cplx* RHS, size_t rhs_stride,
                   cplx* Y, size_t stride, void* Yb, void* Zb, int ld, int m, int kl, int ku) {
    if (prec == 32) return wband_solve_t<cplxf>(h, nf, (cplxf**)dbases, dpvs, dperms, d_perm, RHS, rhs_stride, Y, stride, (cplxf*)Yb, (cplxf*)Zb, ld, m, kl, ku);
    return wband_solve_t<cplx>(h, nf, (cplx**)dbases, dpvs, dperms, d_perm, RHS, rhs_stride, Y, stride, (cplx*)Yb, (cplx*)Zb, ld, m, kl, ku);
}

// =======================================================================================
// Multifrontal sparse LU (the reference's `lu(z*B - A)` on a SparseMatrixCSC is UMFPACK: src/sparse/feast_sparse.jl:334-342).
// Symbolic phase: fh_mf.hpp (nested dissection, fronts, padded groups, maps; pure C++).  Numeric phase, here: the fronts
// of one group -- (fronts of the group) x (quadrature nodes) dense matrices of ONE padded geometry (np fully-summed rows and
// columns, nb boundary rows, order n = np + nb) -- are a batch for the dense LU kernels above:
//   assemble     zero + identity on the pad pivots (k_mf_init), z B - A entries through the plan's assembly list
//                (k_mf_assemble), the children's Schur complements through the extend-add maps (k_mf_extend_add; the two
//                children of a parent one after the other, so the sums are reproducible)
//   partial LU   the two-level right-looking LU above stopped after np columns, pivots searched among the rows < np only
//                (k_lu_panel_reg's pivot limit); what is left in the trailing nb x nb block is the Schur complement
//   store        the L block column (n x np, U11 inside), U12 (np x nb) and the inverted diagonal blocks go to the compact
//                per-node factor store; the full n x n work matrices live in an arena only until their parents are assembled
// Substitution (k_mf_fwd_* / k_mf_bwd_*): right-hand sides travel up the tree as front vectors (n x ld row-major panels, the
// same extend-add maps), each group a batch for k_solve_diag[_inv] / k_solve_update; the solution travels down by gathers.
// No atomics anywhere: bitwise reproducible.  complex128 factors, or complex64 factors (half the store, the f32 MFMA) behind the
// caller's fp64 refinement loop (fh_dense_lu_refined, as for the dense and band factors).
// =======================================================================================
#include "fh_mf.hpp"

struct mf_kid {
    long long child_work, parent_work;     // element offsets of the two front matrices inside one node's work arena
    long long c_rhs_off;                   // rows: the child's group in the substitution panels
    int n_c, np_c, nbnd, rel_off;
    int c_slot, c_F, p_slot, pad;
};
struct mf_slot {                           // per front, in (group, slot) order
    int npiv, nbnd, piv0, rel_off;
    long long p_rhs_off;                   // parent's group in the substitution panels (rows), -1: root
    int p_F, p_n, p_slot, pad;
};

struct fh_mf_state {
    fh_mf::plan P;
    int *d_asm_dst = nullptr, *d_asm_src = nullptr, *d_rel = nullptr, *d_perm = nullptr;
    mf_kid* d_kids = nullptr;
    mf_slot* d_slots = nullptr;
    std::vector<size_t> slot_off;          // group -> first entry of d_slots
    std::vector<size_t> kid_off[2];        // group, side -> first entry of d_kids
    double band_flops = 0.0;
    // the groups of one tree height are independent: with FH_MF_STREAMS=2..4 they run side by side on that many streams (the
    // mid-height groups are 16 - 80 matrices whose panel chains leave most of the chip idle); default: one stream, see fh_mf_make_plan
    hipStream_t extra[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    int nextra = 0;
};

// runs body(g) for every group, level by level (ascending or descending heights), the groups of a level round-robin over the
// main stream and the extra streams; h->stream is switched for the duration of a group (launch helpers and profiler follow it)
template <typename Body>
static int mf_for_levels(feasthip_ctx* h, fh_mf_state* S, bool ascending, Body body) {
    const fh_mf::plan& P = S->P;
    const int ng = (int)P.groups.size();
    std::vector<std::pair<int, int>> levels;
    for (int g = 0; g < ng;) {
        int e = g;
        while (e + 1 < ng && P.groups[e + 1].height == P.groups[g].height) ++e;
        levels.push_back({g, e});
        g = e + 1;
    }
    if (!ascending) std::reverse(levels.begin(), levels.end());
    const hipStream_t main_s = h->stream;
    int rc = 0;
    // Default: the handle's OWN second stream -- the CU-masked side stream of the LU look-ahead (no further hardware queue) --
    // takes every second group of a level.  FH_MF_SIDE=0: strictly one stream.
    hipStream_t extra[3] = {S->extra[0], S->extra[1], S->extra[2]};
    hipEvent_t ev_fork = S->ev_fork, ev_join[3] = {S->ev_join[0], S->ev_join[1], S->ev_join[2]};
    int nextra = S->nextra;
    static const bool side_off = getenv("FH_MF_SIDE") && atoi(getenv("FH_MF_SIDE")) == 0;
    if (nextra == 0 && !side_off && h->lu_lookahead != 0 && lu_side_stream(h, lu_lookahead_reserve(16)) && h->side_stream && h->lu_ev_next && h->lu_ev_rest) {
        extra[0] = h->side_stream; ev_fork = h->lu_ev_next; ev_join[0] = h->lu_ev_rest; nextra = 1;
    }
    for (const auto& lv : levels) {
        const int cnt = lv.second - lv.first + 1;
        const int ns = std::min(cnt, 1 + nextra);
        if (ns > 1) {
            if (hipEventRecord(ev_fork, main_s) != hipSuccess) { h->last_error = "hipEventRecord(multifrontal)"; return FEASTHIP_ERROR_INTERNAL; }
            for (int q = 1; q < ns; ++q)
                if (hipStreamWaitEvent(extra[q - 1], ev_fork, 0) != hipSuccess) { h->last_error = "hipStreamWaitEvent(multifrontal)"; return FEASTHIP_ERROR_INTERNAL; }
        }
        // the largest groups first, one after the other on the streams in turn
        std::vector<int> order(cnt);
        for (int q = 0; q < cnt; ++q) order[q] = lv.first + q;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return P.groups[a].flops > P.groups[b].flops; });
        for (int q = 0; q < cnt && !rc; ++q) {
            const int st = q % ns;
            h->stream = st == 0 ? main_s : extra[st - 1];
            rc = body(order[q]);
        }
        h->stream = main_s;
        if (ns > 1) {
            for (int q = 1; q < ns; ++q) {
                if (hipEventRecord(ev_join[q - 1], extra[q - 1]) != hipSuccess || hipStreamWaitEvent(main_s, ev_join[q - 1], 0) != hipSuccess) {
                    h->last_error = "hipEventRecord(multifrontal join)";
                    return FEASTHIP_ERROR_INTERNAL;
                }
            }
        }
        if (rc) return rc;
    }
    return 0;
}

void fh_mf_free(feasthip_ctx* h) {
    fh_mf_state* S = (fh_mf_state*)h->mf;
    if (!S) return;
    for (void* p : {(void*)S->d_asm_dst, (void*)S->d_asm_src, (void*)S->d_rel, (void*)S->d_perm, (void*)S->d_kids, (void*)S->d_slots}) if (p) (void)hipFree(p);
    for (int q = 0; q < 3; ++q) {
        if (S->extra[q]) { (void)hipStreamSynchronize(S->extra[q]); (void)hipStreamDestroy(S->extra[q]); }
        if (S->ev_join[q]) (void)hipEventDestroy(S->ev_join[q]);
    }
    if (S->ev_fork) (void)hipEventDestroy(S->ev_fork);
    delete S;
    h->mf = nullptr;
}

// Builds the plan from the host copy of the stored pattern.  Returns 0 and leaves h->mf set, or an error code (h->mf null).
int fh_mf_make_plan(feasthip_ctx* h, int leaf) {
    fh_mf_free(h);
    const int N = (int)h->csr.N;
    fh_mf_state* S = new fh_mf_state();
    const int rc = fh_mf::make_plan(N, h->host_rowptr, h->host_col, h->csr.b_identity != 0, leaf, S->P);
    if (rc) { delete S; h->last_error = "multifrontal plan: internal error " + std::to_string(rc); return FEASTHIP_ERROR_INTERNAL; }
    const fh_mf::plan& P = S->P;
    std::vector<mf_slot> slots;
    std::vector<mf_kid> kids;
    const int ng = (int)P.groups.size();
    S->slot_off.assign(ng + 1, 0);
    S->kid_off[0].assign(ng + 1, 0); S->kid_off[1].assign(ng + 1, 0);
    for (int g = 0; g < ng; ++g) {
        const fh_mf::group& G = P.groups[g];
        S->slot_off[g] = slots.size();
        for (int f : G.fronts) {
            const fh_mf::front& F = P.fronts[f];
            mf_slot s;
            s.npiv = F.npiv; s.nbnd = F.nbnd; s.piv0 = F.piv0; s.rel_off = (int)F.bnd_off; s.pad = 0;
            if (F.parent >= 0) {
                const fh_mf::front& Pf = P.fronts[F.parent];
                const fh_mf::group& Gp = P.groups[Pf.group];
                s.p_rhs_off = (long long)Gp.rhs_off; s.p_F = (int)Gp.fronts.size(); s.p_n = Gp.n; s.p_slot = Pf.slot;
            } else { s.p_rhs_off = -1; s.p_F = 0; s.p_n = 0; s.p_slot = 0; }
            slots.push_back(s);
        }
        for (int side = 0; side < 2; ++side) {
            S->kid_off[side][g] = kids.size();
            for (int c : G.kids[side]) {
                const fh_mf::front& C = P.fronts[c];
                const fh_mf::group& Gc = P.groups[C.group];
                const fh_mf::front& Pf = P.fronts[C.parent];
                mf_kid k;
                k.child_work = (long long)(Gc.work_off + (size_t)C.slot * Gc.work_per);
                k.parent_work = (long long)(G.work_off + (size_t)Pf.slot * G.work_per);
                k.c_rhs_off = (long long)Gc.rhs_off;
                k.n_c = Gc.n; k.np_c = Gc.np; k.nbnd = C.nbnd; k.rel_off = (int)C.bnd_off;
                k.c_slot = C.slot; k.c_F = (int)Gc.fronts.size(); k.p_slot = Pf.slot; k.pad = 0;
                kids.push_back(k);
            }
        }
    }
    S->slot_off[ng] = slots.size();
    // (kid_off[side][g + 1] is not the end of side's range: ranges are [kid_off[0][g], kid_off[1][g]) and [kid_off[1][g], next kid_off[0]))
    S->kid_off[0][ng] = kids.size(); S->kid_off[1][ng] = kids.size();
    if (P.bnd.size() > (size_t)0x7fffffff) { delete S; h->last_error = "multifrontal plan: boundary lists beyond int32"; return FEASTHIP_ERROR_MEMORY; }
    auto up = [&](const void* src, size_t bytes, void** dst) -> bool {
        if (bytes == 0) { *dst = nullptr; return true; }
        if (hipMalloc(dst, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    h->mf = S;
    {
        // default ONE stream.  Measured on cfg 3 with four: factorisation 110 -> 101 ms, sweep 24.7 -> 23.6 ms -- and the dense
        // LU on the same handle fell from 0.069 to 0.104 s per cfg-2 solve: with four streams of the handle in use the CU-masked
        // side stream of the LU look-ahead no longer gets a hardware queue of its own and serialises with the main stream.
        const int want = getenv("FH_MF_STREAMS") ? std::max(1, std::min(4, atoi(getenv("FH_MF_STREAMS")))) : 1;
        bool ok = hipEventCreateWithFlags(&S->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int q = 0; q + 1 < want && ok; ++q) {
            ok = hipStreamCreateWithFlags(&S->extra[q], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&S->ev_join[q], hipEventDisableTiming) == hipSuccess;
            if (ok) S->nextra = q + 1;
        }
        if (!ok) (void)hipGetLastError();          // fewer streams (or none): the levels run in order on the main stream
    }
    if (!up(P.asm_dst.data(), P.asm_dst.size() * sizeof(int), (void**)&S->d_asm_dst) || !up(P.asm_src.data(), P.asm_src.size() * sizeof(int), (void**)&S->d_asm_src) ||
        !up(P.rel.data(), P.rel.size() * sizeof(int), (void**)&S->d_rel) || !up(P.perm.data(), P.perm.size() * sizeof(int), (void**)&S->d_perm) ||
        !up(kids.data(), kids.size() * sizeof(mf_kid), (void**)&S->d_kids) || !up(slots.data(), slots.size() * sizeof(mf_slot), (void**)&S->d_slots)) {
        fh_mf_free(h);
        h->last_error = "multifrontal plan: device allocation";
        return FEASTHIP_ERROR_MEMORY;
    }
    return 0;
}
// largest number of fronts in one group: (fronts x nodes of a call) is a grid dimension, the caller batches its nodes accordingly
int fh_mf_max_group(feasthip_ctx* h) {
    if (!h->mf) return 0;
    size_t m = 1;
    for (const fh_mf::group& G : ((fh_mf_state*)h->mf)->P.groups) m = std::max(m, G.fronts.size());
    return (int)m;
}
int fh_mf_max_front(feasthip_ctx* h) { return h->mf ? ((fh_mf_state*)h->mf)->P.max_n : 0; }
double fh_mf_plan_flops(feasthip_ctx* h) { return h->mf ? ((fh_mf_state*)h->mf)->P.flops : 0.0; }
size_t fh_mf_store_bytes(feasthip_ctx* h, int prec) { return h->mf ? ((fh_mf_state*)h->mf)->P.store_elems * (prec == 32 ? sizeof(cplxf) : sizeof(cplx)) : 0; }
size_t fh_mf_pivot_ints(feasthip_ctx* h) { return h->mf ? ((fh_mf_state*)h->mf)->P.piv_ints : 0; }
size_t fh_mf_work_bytes(feasthip_ctx* h, int prec) { return h->mf ? ((fh_mf_state*)h->mf)->P.work_elems * (prec == 32 ? sizeof(cplxf) : sizeof(cplx)) : 0; }

// zero the work matrices of a group, 1 on the pad pivots
template <typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_init(T* const* W, const mf_slot* slots, int F, int n, int np) {
    const int m = blockIdx.y;
    T* A = W[m];
    const int npiv = slots[m % F].npiv;
    const size_t total = (size_t)n * n;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const int r = (int)(e % n), c = (int)(e / n);
        A[e] = LU_MK((r == c && r >= npiv && r < np) ? 1.0 : 0.0, 0.0);
    }
}

// entries of z B - A through the assembly list of one group (grid.y = node)
template <typename VT, bool BIDENT, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_assemble(const int* __restrict__ dst, const int* __restrict__ src, size_t count, T* base, size_t node_stride,
                                                           const VT* __restrict__ aval, const VT* __restrict__ bval, const cplx* z) {
    T* A = base + (size_t)blockIdx.y * node_stride;
    const cplx zz = z[blockIdx.y];
    for (size_t q = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; q < count; q += (size_t)gridDim.x * FH_BLOCK) {
        int d = dst[q];
        const bool dg = d < 0;
        if (dg) d = ~d;
        const int k = src[q];
        cplx v = cmake(0, 0);
        if (k >= 0) {
            if constexpr (sizeof(VT) == sizeof(cplx)) v = cmake(-aval[k].x, -aval[k].y); else v = cmake(-aval[k], 0.0);
            if (!BIDENT) {
                cplx b;
                if constexpr (sizeof(VT) == sizeof(cplx)) b = cmake(bval[k].x, bval[k].y); else b = cmake(bval[k], 0.0);
                cfma(v, zz, b);
            }
        }
        if (dg) v = cadd(v, zz);
        A[d] = cvt<T>(v);
    }
}

// parent front += Schur complement of a child (grid.y = child, grid.z = node)
template <typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_extend_add(const mf_kid* kids, T* base, size_t node_stride, const int* __restrict__ rel, int n_p) {
    const mf_kid kd = kids[blockIdx.y];
    const T* C = base + (size_t)blockIdx.z * node_stride + kd.child_work;
    T* Pm = base + (size_t)blockIdx.z * node_stride + kd.parent_work;
    const int* r = rel + kd.rel_off;
    const int nb = kd.nbnd;
    const size_t total = (size_t)nb * nb;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const int i = (int)(e % nb), j = (int)(e / nb);
        const T v = C[(size_t)(kd.np_c + i) + (size_t)(kd.np_c + j) * kd.n_c];
        T* d = Pm + (size_t)r[i] + (size_t)r[j] * n_p;
        *d = LU_MK(d->x + v.x, d->y + v.y);
    }
}

// L block column (n x np, contiguous) and U12 (np x nb, leading dimension np) of a factored front -> factor store
template <typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_store(T* const* W, T* const* S, int n, int np, int nb, size_t u12_off) {
    const T* A = W[blockIdx.y];
    T* D = S[blockIdx.y];
    const size_t nl = (size_t)n * np, nu = (size_t)np * nb;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < nl + nu; e += (size_t)gridDim.x * FH_BLOCK) {
        if (e < nl) D[e] = A[e];
        else {
            const size_t u = e - nl;
            const int i = (int)(u % np), j = (int)(u / np);
            D[u12_off + u] = A[(size_t)i + (size_t)(np + j) * n];
        }
    }
}

// ---- substitution: front vectors are row-major n x LD panels; group g's panels start at row rhs_off * nf, matrix m = node * F + slot
// forward, step 1: Z[m] = [rhs rows of the pivots (unpermuted); 0]
template <int LD, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_fwd_load(const cplx* __restrict__ RHS, size_t rhs_stride, const int* __restrict__ perm, const mf_slot* slots, int F,
                                                           T* Z, int n) {
    const int m = blockIdx.y, s = m % F, node = m / F;
    const mf_slot sl = slots[s];
    const cplx* R = RHS + (size_t)node * rhs_stride;
    T* z = Z + (size_t)m * n * LD;
    const size_t total = (size_t)n * LD;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const int r = (int)(e / LD), c = (int)(e % LD);
        z[e] = r < sl.npiv ? cvt<T>(R[(size_t)perm[sl.piv0 + r] * LD + c]) : LU_MK(0, 0);
    }
}
// forward, step 2: parent rows += the child's updated boundary rows (grid.y = child, grid.z = node)
template <int LD, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_fwd_add(const mf_kid* kids, const T* __restrict__ Ybase, T* Zp, const int* __restrict__ rel, int nf, int F_p, int n_p) {
    const mf_kid kd = kids[blockIdx.y];
    const int node = blockIdx.z;
    const T* yc = Ybase + ((size_t)kd.c_rhs_off * nf + ((size_t)node * kd.c_F + kd.c_slot) * kd.n_c + kd.np_c) * LD;
    T* zp = Zp + ((size_t)node * F_p + kd.p_slot) * n_p * LD;
    const int* r = rel + kd.rel_off;
    const size_t total = (size_t)kd.nbnd * LD;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const int i = (int)(e / LD), c = (int)(e % LD);
        T* d = zp + (size_t)r[i] * LD + c;
        const T v = yc[e];
        *d = LU_MK(d->x + v.x, d->y + v.y);
    }
}
// forward, step 3: Y[m] = rows of Z[m] in pivot order (rows < np), boundary rows as they are
template <int LD, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_fwd_perm(int* const* pivs, const T* __restrict__ Z, T* Y, int n, int np) {
    const int m = blockIdx.y;
    const int* pr = pivs[m] + np;
    const T* z = Z + (size_t)m * n * LD;
    T* y = Y + (size_t)m * n * LD;
    const size_t total = (size_t)n * LD;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const int r = (int)(e / LD), c = (int)(e % LD);
        y[e] = z[(size_t)(r < np ? pr[r] : r) * LD + c];
    }
}
// backward, step 1: boundary rows of Y[m] = the parent's solution rows (pad rows zero)
template <int LD, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_bwd_load(const mf_slot* slots, int F, const T* Ybase, T* Y, const int* __restrict__ rel, int nf, int n, int np) {
    const int m = blockIdx.y, s = m % F, node = m / F;
    const mf_slot sl = slots[s];
    T* y = Y + ((size_t)m * n + np) * LD;
    const T* yp = sl.p_rhs_off >= 0 ? Ybase + ((size_t)sl.p_rhs_off * nf + ((size_t)node * sl.p_F + sl.p_slot) * sl.p_n) * LD : nullptr;
    const int* r = rel + sl.rel_off;
    const size_t total = (size_t)(n - np) * LD;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const int i = (int)(e / LD), c = (int)(e % LD);
        y[e] = (i < sl.nbnd && yp) ? yp[(size_t)r[i] * LD + c] : LU_MK(0, 0);
    }
}
// backward, last step: the pivots' solution rows go to the caller's panel
template <int LD, typename T>
__global__ __launch_bounds__(FH_BLOCK) void k_mf_scatter(const mf_slot* slots, int F, const T* __restrict__ Y, const int* __restrict__ perm, cplx* OUT, size_t out_stride, int n) {
    const int m = blockIdx.y, s = m % F, node = m / F;
    const mf_slot sl = slots[s];
    const T* y = Y + (size_t)m * n * LD;
    cplx* o = OUT + (size_t)node * out_stride;
    const size_t total = (size_t)sl.npiv * LD;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const int r = (int)(e / LD), c = (int)(e % LD);
        o[(size_t)perm[sl.piv0 + r] * LD + c] = cmake((double)y[e].x, (double)y[e].y);
    }
}

// pointer arrays of one call: per group and matrix (node-major, m = q * F + slot) the work matrix, the factor store, the
// shifted U12 view and the pivots
template <typename T> struct mf_ptrs { T** work; T** store; T** u12; int** piv; std::vector<size_t> off; };
template <typename T>
static int mf_pointer_arrays(feasthip_ctx* h, const fh_mf_state& S, int nf, void* const* stores, int* const* pivs, T* work, bool need_work, mf_ptrs<T>& out) {
    const fh_mf::plan& P = S.P;
    const int ng = (int)P.groups.size();
    out.off.assign(ng + 1, 0);
    for (int g = 0; g < ng; ++g) out.off[g + 1] = out.off[g] + P.groups[g].fronts.size() * (size_t)nf;
    const size_t tot = out.off[ng];
    std::vector<T*> hw(tot), hs(tot), hu(tot);
    std::vector<int*> hp(tot);
    for (int g = 0; g < ng; ++g) {
        const fh_mf::group& G = P.groups[g];
        const size_t F = G.fronts.size();
        const size_t u12 = (size_t)G.n * G.np + fh_mf::inv32_elems(G.np) + (G.inv128 ? fh_mf::inv128_elems(G.np) : 0);
        for (int q = 0; q < nf; ++q)
            for (size_t s = 0; s < F; ++s) {
                const size_t m = out.off[g] + (size_t)q * F + s;
                hw[m] = need_work ? work + (size_t)q * P.work_elems + G.work_off + s * G.work_per : nullptr;
                hs[m] = (T*)stores[q] + G.store_off + s * G.store_per;
                hu[m] = (T*)((uintptr_t)(hs[m] + u12) - (uintptr_t)((size_t)G.np * G.np * sizeof(T)));   // column c >= np of a leading-dimension-np view
                hp[m] = pivs[q] + G.piv_off + s * 2 * (size_t)G.np;
            }
    }
    void* p;
    int rc;
    if ((rc = fh_get_buf(h, "mf_ptrs", 4 * tot * sizeof(void*), &p))) return rc;
    out.work = (T**)p; out.store = out.work + tot; out.u12 = out.store + tot; out.piv = (int**)(out.u12 + tot);
    FH_CHECK(hipMemcpyAsync(out.work, hw.data(), tot * sizeof(void*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(out.store, hs.data(), tot * sizeof(void*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(out.u12, hu.data(), tot * sizeof(void*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(out.piv, hp.data(), tot * sizeof(void*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}

static inline unsigned mf_blocks(size_t work, size_t per_block, unsigned cap) {
    return (unsigned)std::max<size_t>(1, std::min<size_t>(cap, (work + per_block - 1) / per_block));
}

// Factor nf shifted matrices z_q B - A into stores[q] / pivs[q].  info_out[q] != 0: a zero or non-finite pivot in some front.
template <typename T>
static int mf_factor_t(feasthip_ctx* h, int nf, void* const* stores, int* const* pivs, const cplx* dz, std::vector<int>& info_out) {
    fh_mf_state* S = (fh_mf_state*)h->mf;
    if (!S) { h->last_error = "multifrontal LU: no plan"; return FEASTHIP_ERROR_INTERNAL; }
    const fh_mf::plan& P = S->P;
    const int ng = (int)P.groups.size();
    void* p;
    int rc;
    const bool dbg = getenv("FH_DEBUG_TIMING") != nullptr;
    const auto t_in = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count(); };
    if ((rc = fh_get_buf(h, "mf_work", (size_t)nf * P.work_elems * sizeof(T), &p))) return rc;
    T* work = (T*)p;
    const double t_work = since();
    mf_ptrs<T> ptr;
    if ((rc = mf_pointer_arrays<T>(h, *S, nf, stores, pivs, work, true, ptr))) return rc;
    const double t_ptrs = since();
    const size_t tot = ptr.off[ng];
    for (int g = 0; g < ng; ++g)
        if (P.groups[g].fronts.size() * (size_t)nf > 65535) { h->last_error = "multifrontal LU: more than 65535 fronts x nodes in one group"; return FEASTHIP_ERROR_FPM; }
    if ((rc = fh_get_buf(h, "mf_info", tot * sizeof(int), &p))) return rc;
    int* dinfo = (int*)p;
    FH_CHECK(hipMemsetAsync(dinfo, 0, tot * sizeof(int), h->stream));
    static const bool m3_off = getenv("FH_LU_3M") && atoi(getenv("FH_LU_3M")) == 0;
    const bool bid = h->csr.b_identity != 0, cz = h->csr.is_complex != 0;
    auto factor_group = [&](int g) -> int {
        const fh_mf::group& G = P.groups[g];
        const int F = (int)G.fronts.size(), nmat = F * nf, n = G.n, np = G.np, nb = G.nb;
        T** W = ptr.work + ptr.off[g];
        int** PV = ptr.piv + ptr.off[g];
        T** ST = ptr.store + ptr.off[g];
        const lu_geom geom{n, n, (size_t)n * n, 0};
        // ---- assemble
        fh_prof_begin(h, "mf_assemble");
        hipLaunchKernelGGL((k_mf_init<T>), dim3(mf_blocks((size_t)n * n, 8 * FH_BLOCK, 64), nmat), dim3(FH_BLOCK), 0, h->stream, W, S->d_slots + S->slot_off[g], F, n, np);
        {
            const size_t cnt = G.asm_end - G.asm_begin;
            if (cnt) {
                const dim3 grid(mf_blocks(cnt, 4 * FH_BLOCK, 1024), nf), block(FH_BLOCK);
                T* base = work + G.work_off;
                const int* dd = S->d_asm_dst + G.asm_begin;
                const int* ss = S->d_asm_src + G.asm_begin;
                if (cz) {
                    if (bid) hipLaunchKernelGGL((k_mf_assemble<cplx, true, T>), grid, block, 0, h->stream, dd, ss, cnt, base, P.work_elems, (const cplx*)h->csr.aval, (const cplx*)nullptr, dz);
                    else hipLaunchKernelGGL((k_mf_assemble<cplx, false, T>), grid, block, 0, h->stream, dd, ss, cnt, base, P.work_elems, (const cplx*)h->csr.aval, (const cplx*)h->csr.bval, dz);
                } else {
                    if (bid) hipLaunchKernelGGL((k_mf_assemble<double, true, T>), grid, block, 0, h->stream, dd, ss, cnt, base, P.work_elems, (const double*)h->csr.aval, (const double*)nullptr, dz);
                    else hipLaunchKernelGGL((k_mf_assemble<double, false, T>), grid, block, 0, h->stream, dd, ss, cnt, base, P.work_elems, (const double*)h->csr.aval, (const double*)h->csr.bval, dz);
                }
            }
        }
        for (int side = 0; side < 2; ++side) {
            const size_t k0 = S->kid_off[side][g], k1 = side == 0 ? S->kid_off[1][g] : S->kid_off[0][g + 1];
            if (k1 > k0) {
                int nbmax = 0;
                for (int c : G.kids[side]) nbmax = std::max(nbmax, P.fronts[c].nbnd);
                if (nbmax > 0)
                    hipLaunchKernelGGL((k_mf_extend_add<T>), dim3(mf_blocks((size_t)nbmax * nbmax, 8 * FH_BLOCK, 128), (unsigned)(k1 - k0), nf), dim3(FH_BLOCK), 0, h->stream,
                                       S->d_kids + k0, work, P.work_elems, S->d_rel, n);
            }
        }
        fh_prof_end(h);
        // ---- partial LU: np columns, pivots among rows < np
        fh_prof_begin(h, "mf_lu");
        if (h->profiling) h->prof_work["mf_lu"] += fh_mf::partial_lu_flops(n, np) * (double)nmat;
        auto laswp = [&](int p0, int cnt, int a0, int a1, int b0, int b1) {
            const int ncols = (a1 - a0) + (b1 - b0);
            if (ncols <= 0) return;
            hipLaunchKernelGGL((k_lu_laswp<T>), dim3((ncols + FH_BLOCK - 1) / FH_BLOCK, nmat), dim3(FH_BLOCK), 0, h->stream, W, PV, n, p0, cnt, a0, a1, b0, b1);
        };
        auto trsm = [&](int k0, int c0, int c1) {
            if (c1 <= c0) return;
            hipLaunchKernelGGL((k_lu_trsm_mul<LU_NB, T>), dim3((c1 - c0 + FH_BLOCK / LU_NB - 1) / (FH_BLOCK / LU_NB), nmat), dim3(FH_BLOCK), 0, h->stream, W, geom, k0, c0, c1);
        };
        auto gemm = [&](int k0, int kd, int r0, int r1, int c0, int c1) {
            if (r1 <= r0 || c1 <= c0) return;
            const int TR = (r1 - r0 + 63) / 64, TC = (c1 - c0 + 63) / 64;
            const int sw = std::min(8, TC);
            const int nsuper = ((TR + 7) / 8) * ((TC + sw - 1) / sw);
            const int full = ((nsuper + 7) / 8) * 8 * 8 * sw;
            // the XCD-aware super-tile order pays on long trailing updates; on a 3 x 3 tile product of 12 000 fronts its idle
            // workgroups were 95 % of the launch (51 ms for one k = 32 update of the leaf group)
            const int compact = (nmat > 64 || 2 * TR * TC <= full) ? 1 : 0;
            const dim3 grid(compact ? TR * TC : full, nmat);
            if constexpr (sizeof(T) != sizeof(cplx)) hipLaunchKernelGGL((k_lu_gemm<LU_NB, T>), grid, dim3(FH_BLOCK), 0, h->stream, W, n, k0, kd, r0, r1, c0, c1, TR, TC, compact);
            else if (m3_off) hipLaunchKernelGGL((k_lu_gemm_direct<LU_NB, T, false>), grid, dim3(FH_BLOCK), 0, h->stream, W, n, k0, kd, r0, r1, c0, c1, TR, TC, compact);
            else hipLaunchKernelGGL((k_lu_gemm_direct<LU_NB, T, true>), grid, dim3(FH_BLOCK), 0, h->stream, W, n, k0, kd, r0, r1, c0, c1, TR, TC, compact);
        };
        int* ginfo = dinfo + ptr.off[g];
        for (int K0 = 0; K0 < np; K0 += SOLVE_KB) {
            const int Kend = std::min(np, K0 + SOLVE_KB);
            for (int k0 = K0; k0 < Kend; k0 += LU_NB) {
                const int nrows = n - k0;
                const dim3 gg(nmat), bb(LU_PANEL_THREADS);
                if (nrows <= 256) hipLaunchKernelGGL((k_lu_panel_reg<1, 16, T, 256>), gg, dim3(256), 0, h->stream, W, PV, geom, n, k0, LU_NB, ginfo, np);
                else if (nrows <= 512) hipLaunchKernelGGL((k_lu_panel_reg<1, 16, T, 512>), gg, dim3(512), 0, h->stream, W, PV, geom, n, k0, LU_NB, ginfo, np);
                else if (nrows > 8 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<16, 1, T>), gg, bb, 0, h->stream, W, PV, geom, n, k0, LU_NB, ginfo, np);
                else if (nrows <= LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<1, 16, T>), gg, bb, 0, h->stream, W, PV, geom, n, k0, LU_NB, ginfo, np);
                else if (nrows <= 2 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<2, 8, T>), gg, bb, 0, h->stream, W, PV, geom, n, k0, LU_NB, ginfo, np);
                else if (nrows <= 4 * LU_PANEL_THREADS) hipLaunchKernelGGL((k_lu_panel_reg<4, 4, T>), gg, bb, 0, h->stream, W, PV, geom, n, k0, LU_NB, ginfo, np);
                else hipLaunchKernelGGL((k_lu_panel_reg<8, 2, T>), gg, bb, 0, h->stream, W, PV, geom, n, k0, LU_NB, ginfo, np);
                laswp(k0, LU_NB, K0, k0, k0 + LU_NB, Kend);
                if (k0 + LU_NB < Kend) {
                    trsm(k0, k0 + LU_NB, Kend);
                    gemm(k0, LU_NB, k0 + LU_NB, n, k0 + LU_NB, Kend);
                }
            }
            laswp(K0, Kend - K0, 0, K0, Kend, n);
            if (Kend < n) {
                for (int k0 = K0; k0 < Kend; k0 += LU_NB) {
                    trsm(k0, Kend, n);
                    if (k0 + LU_NB < Kend) gemm(k0, LU_NB, k0 + LU_NB, Kend, Kend, n);
                }
                gemm(K0, Kend - K0, Kend, n, Kend, n);
            }
        }
        fh_prof_end(h);
        // ---- store: L block column + U12, inverted diagonal blocks, row permutation of the pivot block
        fh_prof_begin(h, "mf_store");
        const size_t i32 = fh_mf::inv32_elems(np), u12 = (size_t)n * np + i32 + (G.inv128 ? fh_mf::inv128_elems(np) : 0);
        hipLaunchKernelGGL((k_mf_store<T>), dim3(mf_blocks((size_t)n * np + (size_t)np * nb, 8 * FH_BLOCK, 64), nmat), dim3(FH_BLOCK), 0, h->stream, W, ST, n, np, nb, u12);
        const lu_geom gs{n, np, (size_t)n * np, (size_t)n * np + i32};
        hipLaunchKernelGGL((k_lu_invert_diag<LU_NB, T>), dim3(np / LU_NB, nmat), dim3(64), 0, h->stream, ST, gs);
        if (G.inv128) {
            const dim3 gi(SOLVE_KB / 16, (np + SOLVE_KB - 1) / SOLVE_KB, nmat);
            hipLaunchKernelGGL((k_solve_diag<16, false, true, T>), gi, dim3(FH_BLOCK), 0, h->stream, ST, (T*)nullptr, (T*)nullptr, (size_t)0, gs, 0, 0);
            hipLaunchKernelGGL((k_solve_diag<16, true, true, T>), gi, dim3(FH_BLOCK), 0, h->stream, ST, (T*)nullptr, (T*)nullptr, (size_t)0, gs, 0, 0);
        }
        hipLaunchKernelGGL(k_build_perm, dim3(nmat), dim3(FH_BLOCK), (size_t)np * sizeof(int), h->stream, PV, np);
        fh_prof_end(h);
        return 0;
    };
    if ((rc = mf_for_levels(h, S, true, factor_group))) return rc;
    const double t_launch = since();
    std::vector<int> hinfo(tot);
    FH_CHECK(hipMemcpyAsync(hinfo.data(), dinfo, tot * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (dbg) fprintf(stderr, "[feasthip] multifrontal LU: work arena %.1f ms, pointer arrays %.1f ms, launches %.1f ms, drained at %.1f ms\n", t_work, t_ptrs - t_work,
                     t_launch - t_ptrs, since());
    info_out.assign(nf, 0);
    for (int g = 0; g < ng; ++g) {
        const size_t F = P.groups[g].fronts.size();
        for (int q = 0; q < nf; ++q)
            for (size_t s = 0; s < F; ++s) if (hinfo[ptr.off[g] + (size_t)q * F + s] && !info_out[q]) info_out[q] = P.fronts[P.groups[g].fronts[s]].piv0 + 1;
    }
    return 0;
}
// prec: 64 = complex128 factors, 32 = complex64 factors (the caller refines in fp64: fh_dense_lu_refined)
int fh_mf_factor(feasthip_ctx* h, int prec, int nf, void* const* stores, int* const* pivs, const cplx* dz, std::vector<int>& info_out) {
    if (prec == 32) return mf_factor_t<cplxf>(h, nf, stores, pivs, dz, info_out);
    return mf_factor_t<cplx>(h, nf, stores, pivs, dz, info_out);
}

template <int LD, typename T>
static int mf_solve_ld(feasthip_ctx* h, fh_mf_state* S, int nf, const mf_ptrs<T>& ptr, const cplx* RHS, size_t rhs_stride, cplx* OUT, size_t out_stride, int m) {
    const fh_mf::plan& P = S->P;
    const int ng = (int)P.groups.size();
    const int cta = std::max(1, std::min(LD / 16, (m + 15) / 16));
    void* p;
    int rc;
    const size_t rows = P.rhs_rows * (size_t)nf;
    if ((rc = fh_get_buf(h, "mf_y", rows * LD * sizeof(T), &p))) return rc;
    T* Y = (T*)p;
    if ((rc = fh_get_buf(h, "mf_z", rows * LD * sizeof(T), &p))) return rc;
    T* Z = (T*)p;
    fh_prof_begin(h, "mf_solve");
    auto diag = [&](bool upper, T** ST, T* IN, T* OUTp, size_t stride, const lu_geom& gd, int K0, int kb, int nmat, bool inv128) {
        const dim3 grid(cta, nmat), block(FH_BLOCK);
        if (inv128) {
            if (upper) hipLaunchKernelGGL((k_solve_diag_inv<LD, true, T>), grid, block, 0, h->stream, ST, IN, OUTp, stride, gd, K0, kb);
            else hipLaunchKernelGGL((k_solve_diag_inv<LD, false, T>), grid, block, 0, h->stream, ST, IN, OUTp, stride, gd, K0, kb);
        } else {
            if (upper) hipLaunchKernelGGL((k_solve_diag<LD, true, false, T>), grid, block, 0, h->stream, ST, IN, OUTp, stride, gd, K0, kb);
            else hipLaunchKernelGGL((k_solve_diag<LD, false, false, T>), grid, block, 0, h->stream, ST, IN, OUTp, stride, gd, K0, kb);
        }
    };
    auto forward_group = [&](int g) -> int {                     // forward: leaves first
        const fh_mf::group& G = P.groups[g];
        const int F = (int)G.fronts.size(), nmat = F * nf, n = G.n, np = G.np;
        T* Yg = Y + G.rhs_off * (size_t)nf * LD;
        T* Zg = Z + G.rhs_off * (size_t)nf * LD;
        const size_t stride = (size_t)n * LD;
        T** ST = ptr.store + ptr.off[g];
        const unsigned gb = mf_blocks((size_t)n * LD, 4 * FH_BLOCK, 64);
        hipLaunchKernelGGL((k_mf_fwd_load<LD, T>), dim3(gb, nmat), dim3(FH_BLOCK), 0, h->stream, RHS, rhs_stride, S->d_perm, S->d_slots + S->slot_off[g], F, Zg, n);
        for (int side = 0; side < 2; ++side) {
            const size_t k0 = S->kid_off[side][g], k1 = side == 0 ? S->kid_off[1][g] : S->kid_off[0][g + 1];
            if (k1 > k0) {
                int nbmax = 0;
                for (int c : G.kids[side]) nbmax = std::max(nbmax, P.fronts[c].nbnd);
                if (nbmax > 0)
                    hipLaunchKernelGGL((k_mf_fwd_add<LD, T>), dim3(mf_blocks((size_t)nbmax * LD, 4 * FH_BLOCK, 64), (unsigned)(k1 - k0), nf), dim3(FH_BLOCK), 0, h->stream,
                                       S->d_kids + k0, Y, Zg, S->d_rel, nf, F, n);
            }
        }
        hipLaunchKernelGGL((k_mf_fwd_perm<LD, T>), dim3(gb, nmat), dim3(FH_BLOCK), 0, h->stream, ptr.piv + ptr.off[g], Zg, Yg, n, np);
        const size_t i32 = fh_mf::inv32_elems(np);
        const lu_geom gd{n, np, (size_t)n * np, (size_t)n * np + i32};      // diagonal solves: bounded by the pivot block
        const lu_geom gu{n, n, (size_t)n * np, (size_t)n * np + i32};       // updates: all rows of the front
        for (int K0 = 0; K0 < np; K0 += SOLVE_KB) {
            const int kb = std::min(SOLVE_KB / LU_NB, (np - K0) / LU_NB);
            const int r0 = K0 + LU_NB * kb;
            diag(false, ST, Yg, Zg, stride, gd, K0, kb, nmat, G.inv128 != 0);
            if (r0 < n)
                hipLaunchKernelGGL((k_solve_update<LD, T>), dim3((n - r0 + 63) / 64, nmat), dim3(FH_BLOCK), 0, h->stream, ST, Yg, Zg, stride, gu, K0, LU_NB * kb, r0, n, cta);
        }
        return 0;
    };
    auto backward_group = [&](int g) -> int {                    // backward: root first
        const fh_mf::group& G = P.groups[g];
        const int F = (int)G.fronts.size(), nmat = F * nf, n = G.n, np = G.np, nb = G.nb;
        T* Yg = Y + G.rhs_off * (size_t)nf * LD;
        T* Zg = Z + G.rhs_off * (size_t)nf * LD;
        const size_t stride = (size_t)n * LD;
        T** ST = ptr.store + ptr.off[g];
        const size_t i32 = fh_mf::inv32_elems(np);
        const lu_geom gd{n, np, (size_t)n * np, (size_t)n * np + i32};
        const lu_geom gu{n, n, (size_t)n * np, (size_t)n * np + i32};
        if (nb > 0) {
            hipLaunchKernelGGL((k_mf_bwd_load<LD, T>), dim3(mf_blocks((size_t)nb * LD, 4 * FH_BLOCK, 64), nmat), dim3(FH_BLOCK), 0, h->stream, S->d_slots + S->slot_off[g], F,
                               Y, Yg, S->d_rel, nf, n, np);
            // z1 -= U12 x2: U12 through its leading-dimension-np view (columns np .. n)
            const lu_geom g12{np, n, 0, 0};
            hipLaunchKernelGGL((k_solve_update<LD, T>), dim3((np + 63) / 64, nmat), dim3(FH_BLOCK), 0, h->stream, ptr.u12 + ptr.off[g], Zg, Yg, stride, g12, np, nb, 0, np, cta);
        }
        for (int K0 = ((np - 1) / SOLVE_KB) * SOLVE_KB; K0 >= 0; K0 -= SOLVE_KB) {
            const int kb = std::min(SOLVE_KB / LU_NB, (np - K0) / LU_NB);
            diag(true, ST, Zg, Yg, stride, gd, K0, kb, nmat, G.inv128 != 0);
            if (K0 > 0)
                hipLaunchKernelGGL((k_solve_update<LD, T>), dim3((K0 + 63) / 64, nmat), dim3(FH_BLOCK), 0, h->stream, ST, Zg, Yg, stride, gu, K0, LU_NB * kb, 0, K0, cta);
        }
        hipLaunchKernelGGL((k_mf_scatter<LD, T>), dim3(mf_blocks((size_t)np * LD, 4 * FH_BLOCK, 64), nmat), dim3(FH_BLOCK), 0, h->stream, S->d_slots + S->slot_off[g], F, Yg,
                           S->d_perm, OUT, out_stride, n);
        return 0;
    };
    if ((rc = mf_for_levels(h, S, true, forward_group))) return rc;
    if ((rc = mf_for_levels(h, S, false, backward_group))) return rc;
    fh_prof_end(h);
    return 0;
}

// OUT[q] = (z_q B - A)^-1 RHS[q] with the factors of fh_mf_factor.  Panels row-major N x ld (fp64 in and out: complex64 factors narrow
// the right-hand side on the way into the front vectors and widen the solution on the way out); rhs_stride = 0: one shared panel.
template <typename T>
static int mf_solve_t(feasthip_ctx* h, int nf, void* const* stores, int* const* pivs, const cplx* RHS, size_t rhs_stride, cplx* OUT, size_t out_stride, int ld, int m) {
    fh_mf_state* S = (fh_mf_state*)h->mf;
    if (!S) { h->last_error = "multifrontal LU: no plan"; return FEASTHIP_ERROR_INTERNAL; }
    mf_ptrs<T> ptr;
    int rc;
    const auto t_in = std::chrono::steady_clock::now();
    if ((rc = mf_pointer_arrays<T>(h, *S, nf, stores, pivs, nullptr, false, ptr))) return rc;
    struct report { std::chrono::steady_clock::time_point t0, t1; ~report() { if (getenv("FH_DEBUG_TIMING")) fprintf(stderr, "[feasthip] multifrontal solve: pointer arrays %.1f ms, buffers + launches %.1f ms (host)\n",
        std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count()); } } rep{t_in, std::chrono::steady_clock::now()};
    if (ld == 16) return mf_solve_ld<16, T>(h, S, nf, ptr, RHS, rhs_stride, OUT, out_stride, m);
    if (ld == 32) return mf_solve_ld<32, T>(h, S, nf, ptr, RHS, rhs_stride, OUT, out_stride, m);
    return mf_solve_ld<64, T>(h, S, nf, ptr, RHS, rhs_stride, OUT, out_stride, m);
}
int fh_mf_solve(feasthip_ctx* h, int prec, int nf, void* const* stores, int* const* pivs, const cplx* RHS, size_t rhs_stride, cplx* OUT, size_t out_stride, int ld, int m) {
    if (prec == 32) return mf_solve_t<cplxf>(h, nf, stores, pivs, RHS, rhs_stride, OUT, out_stride, ld, m);
    return mf_solve_t<cplx>(h, nf, stores, pivs, RHS, rhs_stride, OUT, out_stride, ld, m);
}
