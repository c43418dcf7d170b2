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

template <typename VT, int LD>
static void launch_dense_op_ld(const fh_dense_op_args& a, int nblk, hipStream_t st) {
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
// batched LU
// ---------------------------------------------------------------------------------------
template <typename VT, bool BIDENT>
__global__ __launch_bounds__(FH_BLOCK) void k_form_shifted(const VT* __restrict__ A, const VT* __restrict__ B,
                                                            cplx* const* LUs, const cplx* z, int N) {
    cplx* S = LUs[blockIdx.y];
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
        S[e] = v;
    }
}

// Panel factorisation: one workgroup per matrix walks the nb panel columns; pivot rule is
// LAPACK's IZAMAX (max |re|+|im|, lowest index on ties).
__global__ __launch_bounds__(LU_PANEL_THREADS) void k_lu_panel(cplx* const* LUs, int* const* pivs, int N, int k0,
                                                                int nb, int* info) {
    cplx* A = LUs[blockIdx.x];
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
            cplx v = A[(size_t)jj * N + i];
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
            cplx u = A[(size_t)(k0 + t) * N + jj];
            A[(size_t)(k0 + t) * N + jj] = A[(size_t)(k0 + t) * N + p];
            A[(size_t)(k0 + t) * N + p] = u;
        }
        __syncthreads();
        const cplx pv = A[(size_t)jj * N + jj];
        const bool singular = (pv.x == 0.0 && pv.y == 0.0);
        const cplx inv = singular ? cmake(0, 0) : cdiv(cmake(1, 0), pv);
        for (int i = jj + 1 + t; i < N; i += LU_PANEL_THREADS) A[(size_t)jj * N + i] = cmul(A[(size_t)jj * N + i], inv);
        __syncthreads();
        const int ncols = k0 + nb - 1 - jj, nrows = N - 1 - jj;
        for (int e = t; e < nrows * ncols; e += LU_PANEL_THREADS) {
            int i = jj + 1 + e % nrows, c = jj + 1 + e / nrows;
            cplx l = A[(size_t)jj * N + i], u = A[(size_t)c * N + jj];
            cplx v = A[(size_t)c * N + i];
            A[(size_t)c * N + i] = csub(v, cmul(l, u));
        }
        __syncthreads();
    }
}

// apply the panel's row interchanges to the columns outside the panel
__global__ __launch_bounds__(FH_BLOCK) void k_lu_laswp(cplx* const* LUs, int* const* pivs, int N, int k0, int nb) {
    cplx* A = LUs[blockIdx.y];
    const int* piv = pivs[blockIdx.y];
    const int c = blockIdx.x * FH_BLOCK + threadIdx.x;
    if (c >= N || (c >= k0 && c < k0 + nb)) return;
    for (int j = 0; j < nb; ++j) {
        int p = piv[k0 + j];
        if (p != k0 + j) {
            cplx u = A[(size_t)c * N + k0 + j];
            A[(size_t)c * N + k0 + j] = A[(size_t)c * N + p];
            A[(size_t)c * N + p] = u;
        }
    }
}

// U12 = L11^{-1} A12 : one thread per trailing column
template <int NB>
__global__ __launch_bounds__(FH_BLOCK) void k_lu_trsm(cplx* const* LUs, int N, int k0) {
    cplx* A = LUs[blockIdx.y];
    __shared__ cplx L[NB][NB + 1];
    for (int e = threadIdx.x; e < NB * NB; e += FH_BLOCK) {
        int i = e % NB, j = e / NB;
        L[i][j] = A[(size_t)(k0 + j) * N + k0 + i];
    }
    __syncthreads();
    const int c = k0 + NB + blockIdx.x * FH_BLOCK + threadIdx.x;
    if (c >= N) return;
    cplx x[NB];
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

// A22 -= L21 U12 on v_mfma_f64_16x16x4_f64: 64x64 tile of A22 per workgroup, four waves as
// 2x2 of 32x32, each wave 2x2 MFMA tiles.  The product is formed TRANSPOSED,
//     D[c][i] = sum_k U[k][c] * L[i][k]      (A operand = U^T, B operand = L^T)
// so that a lane's results (row = (l>>4)+4r -> c, col = l&15 -> i) are 16 consecutive rows of
// one column of the column-major trailing matrix: 256 B contiguous per 16-lane group.
// Complex product from four real MFMAs per tile: rr, ii, ri, ir;  Re = rr - ii, Im = ri + ir.
typedef double lu_v4d __attribute__((ext_vector_type(4)));

template <int NB>
__global__ __launch_bounds__(FH_BLOCK) void k_lu_gemm(cplx* const* LUs, int N, int k0) {
    cplx* A = LUs[blockIdx.z];
    __shared__ cplx Ls[NB][64];
    __shared__ cplx Us[NB][64];
    const int t = threadIdx.x;
    const int base = k0 + NB;
    const int i0 = base + blockIdx.x * 64, c0 = base + blockIdx.y * 64;
    const int lane = t & 63, wave = t >> 6;
    const int wi = (wave & 1) * 32, wc = (wave >> 1) * 32;     // wave's 32x32 sub-tile
    const int lr = lane & 15, lk = lane >> 4;
    // issue the read of this lane's 16 trailing-matrix entries first: it overlaps the LDS fill
    // and the MFMA loop instead of sitting exposed in front of the read-modify-write
    cplx cv[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + wc + 16 * a + lk + 4 * r;
                const int i = i0 + wi + 16 * b + lr;
                cv[a][b][r] = (i < N && c < N) ? A[(size_t)c * N + i] : cmake(0, 0);
            }
    for (int e = t; e < NB * 64; e += FH_BLOCK) {
        int ii = e % 64, k = e / 64;
        Ls[k][ii] = (i0 + ii < N) ? A[(size_t)(k0 + k) * N + i0 + ii] : cmake(0, 0);
    }
    for (int e = t; e < NB * 64; e += FH_BLOCK) {
        int k = e % NB, cc = e / NB;
        Us[k][cc] = (c0 + cc < N) ? A[(size_t)(c0 + cc) * N + k0 + k] : cmake(0, 0);
    }
    __syncthreads();
    lu_v4d rr[2][2], ii[2][2], ri[2][2], ir[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { rr[a][b] = (lu_v4d){0, 0, 0, 0}; ii[a][b] = rr[a][b]; ri[a][b] = rr[a][b]; ir[a][b] = rr[a][b]; }
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
        cplx u[2], l[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) u[a] = Us[kk + lk][wc + 16 * a + lr];
#pragma unroll
        for (int b = 0; b < 2; ++b) l[b] = Ls[kk + lk][wi + 16 * b + lr];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                rr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[a].x, l[b].x, rr[a][b], 0, 0, 0);
                ii[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[a].y, l[b].y, ii[a][b], 0, 0, 0);
                ri[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[a].x, l[b].y, ri[a][b], 0, 0, 0);
                ir[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[a].y, l[b].x, ir[a][b], 0, 0, 0);
            }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + wc + 16 * a + lk + 4 * r;
                const int i = i0 + wi + 16 * b + lr;
                if (i < N && c < N) {
                    cplx v = cv[a][b][r];
                    v.x -= rr[a][b][r] - ii[a][b][r];
                    v.y -= ri[a][b][r] + ir[a][b][r];
                    A[(size_t)c * N + i] = v;
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
template <int NB>
__global__ __launch_bounds__(64) void k_lu_invert_diag(cplx* const* LUs, int N) {
    cplx* A = LUs[blockIdx.y];
    const int k0 = blockIdx.x * NB;
    const int nb = min(NB, N - k0);
    cplx* inv = A + (size_t)N * N + (size_t)blockIdx.x * 2 * NB * NB;
    __shared__ cplx T[NB][NB + 1];
    __shared__ cplx Li[NB][NB + 1];
    __shared__ cplx Ui[NB][NB + 1];
    const int t = threadIdx.x;
    for (int e = t; e < NB * NB; e += 64) {
        int i = e % NB, j = e / NB;
        T[i][j] = (i < nb && j < nb) ? A[(size_t)(k0 + j) * N + k0 + i] : cmake(i == j ? 1.0 : 0.0, 0.0);
        Li[i][j] = cmake(0, 0);
        Ui[i][j] = cmake(0, 0);
    }
    __syncthreads();
    if (t < NB) {                       // column t of L^-1
        const int c = t;
        Li[c][c] = cmake(1, 0);
        for (int i = c + 1; i < NB; ++i) {
            cplx s = cmake(0, 0);
            for (int j = c; j < i; ++j) cfma(s, T[i][j], Li[j][c]);
            Li[i][c] = cmake(-s.x, -s.y);
        }
    } else if (t < 2 * NB) {            // column c of U^-1
        const int c = t - NB;
        Ui[c][c] = cdiv(cmake(1, 0), T[c][c]);
        for (int i = c - 1; i >= 0; --i) {
            cplx s = cmake(0, 0);
            for (int j = i + 1; j <= c; ++j) cfma(s, T[i][j], Ui[j][c]);
            Ui[i][c] = cdiv(cmake(-s.x, -s.y), T[i][i]);
        }
    }
    __syncthreads();
    for (int e = t; e < NB * NB; e += 64) {
        int i = e % NB, j = e / NB;
        inv[e] = Li[i][j];
        inv[NB * NB + e] = Ui[i][j];
    }
}

// Y[node][i,:] = RHS[perm[i],:]
__global__ __launch_bounds__(FH_BLOCK) void k_gather_rows(const cplx* __restrict__ RHS, int* const* perms,
                                                           cplx* __restrict__ Y, size_t stride, int N, int ld) {
    const int* perm = perms[blockIdx.y];
    cplx* Yn = Y + (size_t)blockIdx.y * stride;
    const size_t total = (size_t)N * ld;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        size_t i = e / ld, c = e % ld;
        Yn[e] = RHS[(size_t)perm[i] * ld + c];
    }
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
template <int NB, int LD, bool UPPER>
__global__ __launch_bounds__(FH_BLOCK) void k_solve_step(cplx* const* LUs, cplx* IN, cplx* OUT, size_t stride, int N,
                                                          int k0, int r0, int r1) {
    static_assert(NB == 32, "tile mapping assumes NB == 32");
    const cplx* A = LUs[blockIdx.y];
    const cplx* inv = A + (size_t)N * N + ((size_t)(k0 / NB) * 2 + (UPPER ? 1 : 0)) * NB * NB;
    cplx* in = IN + (size_t)blockIdx.y * stride;
    cplx* out = OUT + (size_t)blockIdx.y * stride;
    __shared__ cplx Zs[NB][LD + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int CT = LD / 16;
    // ---- slab product
    for (int q = wave; q < 2 * CT; q += 4) {
        const int ti = q / CT, ta = q % CT;
        lu_v4d re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < NB; kk += 4) {
            const cplx a = inv[(size_t)(kk + lk) * NB + 16 * ti + lr];
            const int row = k0 + kk + lk;
            const cplx b = row < N ? in[(size_t)row * LD + 16 * ta + lr] : cmake(0, 0);
            re = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, re, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, b.y, re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, im, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, im, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + lk + 4 * r, c = 16 * ta + lr;
            const cplx z = cmake(re[r], im[r]);
            Zs[i][c] = z;
            if (blockIdx.x == 0 && k0 + i < N) out[(size_t)(k0 + i) * LD + c] = z;
        }
    }
    __syncthreads();
    // ---- update of this wave's 16-row band
    const int ib = r0 + blockIdx.x * 64 + 16 * wave;
    if (ib >= r1) return;
    cplx am[NB / 4];
#pragma unroll
    for (int s = 0; s < NB / 4; ++s) {
        const int col = k0 + 4 * s + lk;
        am[s] = (ib + lr < r1 && col < N) ? A[(size_t)col * N + ib + lr] : cmake(0, 0);
    }
#pragma unroll
    for (int ta = 0; ta < CT; ++ta) {
        cplx y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ib + lk + 4 * r;
            y[r] = i < r1 ? in[(size_t)i * LD + 16 * ta + lr] : cmake(0, 0);
        }
        lu_v4d re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NB / 4; ++s) {
            const cplx b = Zs[4 * s + lk][16 * ta + lr];
            re = __builtin_amdgcn_mfma_f64_16x16x4f64(am[s].x, b.x, re, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f64_16x16x4f64(-am[s].y, b.y, re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f64_16x16x4f64(am[s].x, b.y, im, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f64_16x16x4f64(am[s].y, b.x, im, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ib + lk + 4 * r;
            if (i < r1) in[(size_t)i * LD + 16 * ta + lr] = cmake(y[r].x - re[r], y[r].y - im[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------
static int lu_factor_batch(feasthip_ctx* h, const std::vector<int>& which, const std::vector<cplx>& zlist,
                           std::vector<int>& info_out) {
    // which: local node slots to (re)factor; zlist: their shifts
    const int nf = (int)which.size();
    if (nf == 0) return 0;
    const int N = (int)h->dense.N;
    void* p;
    int rc;
    std::vector<cplx*> lus(nf);
    std::vector<int*> pvs(nf);
    for (int q = 0; q < nf; ++q) { lus[q] = (cplx*)h->lu_factors[which[q]]; pvs[q] = h->lu_pivots[which[q]]; }
    if ((rc = fh_get_buf(h, "lu_ptrs", nf * sizeof(cplx*), &p))) return rc;
    cplx** dlus = (cplx**)p;
    if ((rc = fh_get_buf(h, "lu_pptrs", nf * sizeof(int*), &p))) return rc;
    int** dpvs = (int**)p;
    if ((rc = fh_get_buf(h, "lu_z", nf * sizeof(cplx), &p))) return rc;
    cplx* dz = (cplx*)p;
    if ((rc = fh_get_buf(h, "lu_info", nf * sizeof(int), &p))) return rc;
    int* dinfo = (int*)p;
    FH_CHECK(hipMemcpyAsync(dlus, lus.data(), nf * sizeof(cplx*), hipMemcpyHostToDevice, h->stream));
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
            if (bid) hipLaunchKernelGGL((k_form_shifted<cplx, true>), grid, block, 0, h->stream, (const cplx*)h->dense.A, (const cplx*)nullptr, dlus, dz, N);
            else hipLaunchKernelGGL((k_form_shifted<cplx, false>), grid, block, 0, h->stream, (const cplx*)h->dense.A, (const cplx*)h->dense.B, dlus, dz, N);
        } else {
            if (bid) hipLaunchKernelGGL((k_form_shifted<double, true>), grid, block, 0, h->stream, (const double*)h->dense.A, (const double*)nullptr, dlus, dz, N);
            else hipLaunchKernelGGL((k_form_shifted<double, false>), grid, block, 0, h->stream, (const double*)h->dense.A, (const double*)h->dense.B, dlus, dz, N);
        }
        fh_prof_end(h);
    }
    for (int k0 = 0; k0 < N; k0 += LU_NB) {
        const int nb = std::min(LU_NB, N - k0);
        fh_prof_begin(h, "lu_panel");
        hipLaunchKernelGGL(k_lu_panel, dim3(nf), dim3(LU_PANEL_THREADS), 0, h->stream, dlus, dpvs, N, k0, nb, dinfo);
        fh_prof_end(h);
        fh_prof_begin(h, "lu_laswp");
        hipLaunchKernelGGL(k_lu_laswp, dim3((N + FH_BLOCK - 1) / FH_BLOCK, nf), dim3(FH_BLOCK), 0, h->stream, dlus, dpvs, N, k0, nb);
        fh_prof_end(h);
        const int rest = N - k0 - nb;
        if (rest > 0) {
            // nb == LU_NB here (a short last panel has no trailing matrix)
            fh_prof_begin(h, "lu_trsm");
            hipLaunchKernelGGL((k_lu_trsm<LU_NB>), dim3((rest + FH_BLOCK - 1) / FH_BLOCK, nf), dim3(FH_BLOCK), 0, h->stream, dlus, N, k0);
            fh_prof_end(h);
            const int tiles = (rest + 63) / 64;
            fh_prof_begin(h, "lu_gemm");
            hipLaunchKernelGGL((k_lu_gemm<LU_NB>), dim3(tiles, tiles, nf), dim3(FH_BLOCK), 0, h->stream, dlus, N, k0);
            fh_prof_end(h);
        }
    }
    fh_prof_begin(h, "lu_invert");
    hipLaunchKernelGGL((k_lu_invert_diag<LU_NB>), dim3((N + LU_NB - 1) / LU_NB, nf), dim3(64), 0, h->stream, dlus, N);
    if (N <= 16000) hipLaunchKernelGGL(k_build_perm, dim3(nf), dim3(FH_BLOCK), (size_t)N * sizeof(int), h->stream, dpvs, N);
    else hipLaunchKernelGGL(k_build_perm_global, dim3(nf), dim3(64), 0, h->stream, dpvs, N);
    fh_prof_end(h);
    info_out.assign(nf, 0);
    FH_CHECK(hipMemcpyAsync(info_out.data(), dinfo, nf * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}

template <int LD>
static void lu_solve_launch(feasthip_ctx* h, cplx** dlus, cplx* Y, cplx* Z, size_t stride, int N, int nf) {
    const int nblocks = (N + LU_NB - 1) / LU_NB;
    for (int b = 0; b < nblocks; ++b) {        // forward: L z = P b   (Y -> Z)
        const int k0 = b * LU_NB, r0 = std::min(N, k0 + LU_NB);
        const int gx = std::max(1, (N - r0 + 63) / 64);
        hipLaunchKernelGGL((k_solve_step<LU_NB, LD, false>), dim3(gx, nf), dim3(FH_BLOCK), 0, h->stream, dlus, Y, Z, stride, N, k0, r0, N);
    }
    for (int b = nblocks - 1; b >= 0; --b) {   // backward: U x = z   (Z -> Y)
        const int k0 = b * LU_NB;
        const int gx = std::max(1, (k0 + 63) / 64);
        hipLaunchKernelGGL((k_solve_step<LU_NB, LD, true>), dim3(gx, nf), dim3(FH_BLOCK), 0, h->stream, dlus, Z, Y, stride, N, k0, 0, k0);
    }
}

static int lu_solve_batch(feasthip_ctx* h, int ld, const std::vector<int>& slots, const cplx* RHS, cplx* Y, size_t stride) {
    const int nf = (int)slots.size();
    const int N = (int)h->dense.N;
    void* p;
    int rc;
    std::vector<cplx*> lus(nf);
    std::vector<int*> perms(nf);
    for (int q = 0; q < nf; ++q) { lus[q] = (cplx*)h->lu_factors[slots[q]]; perms[q] = h->lu_pivots[slots[q]] + N; }
    if ((rc = fh_get_buf(h, "lu_ptrs", nf * sizeof(cplx*), &p))) return rc;
    cplx** dlus = (cplx**)p;
    if ((rc = fh_get_buf(h, "lu_permptrs", nf * sizeof(int*), &p))) return rc;
    int** dperms = (int**)p;
    if ((rc = fh_get_buf(h, "lu_zpanel", (size_t)nf * stride * sizeof(cplx), &p))) return rc;
    cplx* Z = (cplx*)p;
    FH_CHECK(hipMemcpyAsync(dlus, lus.data(), nf * sizeof(cplx*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipMemcpyAsync(dperms, perms.data(), nf * sizeof(int*), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    fh_prof_begin(h, "lu_solve");
    hipLaunchKernelGGL(k_gather_rows, dim3(fh_vec_nblk(N, ld), nf), dim3(FH_BLOCK), 0, h->stream, RHS, dperms, Y, stride, N, ld);
    if (ld == 16) lu_solve_launch<16>(h, dlus, Y, Z, stride, N, nf);
    else if (ld == 32) lu_solve_launch<32>(h, dlus, Y, Z, stride, N, nf);
    else lu_solve_launch<64>(h, dlus, Y, Z, stride, N, nf);
    fh_prof_end(h);
    return 0;
}

static int lu_ensure_slots(feasthip_ctx* h, int nslots) {
    const size_t N = (size_t)h->dense.N;
    while ((int)h->lu_factors.size() < nslots) {
        void* f = nullptr; int* pv = nullptr;
        // factor, then the inverted diagonal blocks (k_lu_invert_diag); pivots, then the row permutation
        const size_t nblk = (N + LU_NB - 1) / LU_NB;
        FH_CHECK(hipMalloc(&f, (N * N + nblk * 2 * LU_NB * LU_NB) * sizeof(cplx)));
        hipError_t e = hipMalloc((void**)&pv, 2 * N * sizeof(int));
        if (e != hipSuccess) { hipFree(f); h->last_error = "hipMalloc(pivots)"; return FEASTHIP_ERROR_MEMORY; }
        h->lu_factors.push_back(f); h->lu_pivots.push_back(pv); h->lu_valid.push_back(0); h->lu_z.push_back(cmake(0, 0));
    }
    return 0;
}

int fh_dense_lu_solve_nodes(feasthip_ctx* h, int ld, int m, int nodes, const std::vector<cplx>& z, const cplx* RHS,
                            cplx* Y, size_t stride, std::vector<int>& status, int64_t* nfact) {
    (void)m;
    int rc = lu_ensure_slots(h, nodes);
    if (rc) return rc;
    std::vector<int> need;
    std::vector<cplx> zl;
    for (int e = 0; e < nodes; ++e) {
        bool ok = h->cache_factors && h->lu_valid[e] == 1 && h->lu_z[e].x == z[e].x && h->lu_z[e].y == z[e].y;
        if (!ok) { need.push_back(e); zl.push_back(z[e]); h->lu_valid[e] = 0; }
    }
    std::vector<int> info;
    if ((rc = lu_factor_batch(h, need, zl, info))) return rc;
    for (size_t q = 0; q < need.size(); ++q) {
        h->lu_z[need[q]] = zl[q];
        h->lu_valid[need[q]] = info[q] == 0 ? 1 : -1;   // -1: singular
    }
    if (nfact) *nfact = (int64_t)need.size();
    std::vector<int> slots(nodes);
    for (int e = 0; e < nodes; ++e) slots[e] = e;
    if ((rc = lu_solve_batch(h, ld, slots, RHS, Y, stride))) return rc;
    status.assign(nodes, 0);
    for (int e = 0; e < nodes; ++e) if (h->lu_valid[e] != 1) status[e] = FEASTHIP_ERROR_LAPACK;
    return 0;
}

int fh_dense_lu_solve_single(feasthip_ctx* h, int ld, int m, cplx z, const cplx* RHS, cplx* Y, int* status, int64_t* nfact) {
    (void)m;
    // uses a dedicated extra slot after the node slots
    const int slot = h->node_count;
    int rc = lu_ensure_slots(h, slot + 1);
    if (rc) return rc;
    std::vector<int> need(1, slot), info;
    std::vector<cplx> zl(1, z);
    bool cached = h->cache_factors && h->lu_valid[slot] == 1 && h->lu_z[slot].x == z.x && h->lu_z[slot].y == z.y;
    if (!cached) {
        if ((rc = lu_factor_batch(h, need, zl, info))) return rc;
        h->lu_z[slot] = z;
        h->lu_valid[slot] = info[0] == 0 ? 1 : -1;
        if (nfact) *nfact = 1;
    }
    if ((rc = lu_solve_batch(h, ld, need, RHS, Y, (size_t)h->dense.N * ld))) return rc;
    *status = h->lu_valid[slot] == 1 ? 0 : FEASTHIP_ERROR_LAPACK;
    return 0;
}
