// fh_sparse.hip -- batched sparse shifted solves for the FEAST contour sweep (gfx950).
//
// Replaces, for all RHS columns and all local quadrature nodes at once:
//   SparseShiftedOperator.mul!            src/sparse/feast_sparse.jl:20-27   (2 SpMV + 2 axpy)
//   solve_shifted_iterative!              src/sparse/feast_sparse.jl:164-203 (per-column GMRES)
//   Krylov.bicgstab matrix-free solver    src/interfaces/feast_matfree.jl:716-718
//
// Data layout: block vectors are row-major N x LD panels (fh_common.hpp) of complex128 or --
// for the mixed-precision correction solves -- complex64; the node batch is walked inside the
// SpMM and is the y dimension of the vector kernels.  The operator is never materialised per
// node: S_c = coefB[c]*B + coefA[c]*A is formed on the fly from the real (or complex) CSR
// values of A and B on their union pattern, with a per-COLUMN complex coefficient pair so
// the same kernel serves (z_e B - A)X, A X, B X and the residual A X - B X diag(lambda).
// Every reduction (dots, norms) accumulates in fp64 whatever the panel precision.
#include "fh_common.hpp"
#include "fh_kernels.hpp"

#define FH_BLOCK 256
#define FH_FIN_BLOCK 1024

template <int LD>
__device__ __forceinline__ void fh_block_reduce_cols(cplx v, cplx* red, cplx* out) {
    // 256 threads; thread t owns column t % LD.  Sum the 256/LD partials of each column.
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    if (t < LD) {
        cplx s = red[t];
#pragma unroll
        for (int k = 1; k < FH_BLOCK / LD; ++k) s = cadd(s, red[t + k * LD]);
        out[t] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------
// SpMM:  Y[node] = (Bvec -) (coefB*B + coefA*A) X[node]      with optional fused dots
//   dot_mode 0: none
//            1: partial1 = <U, Y>                 (BiCGStab sigma = <rhat, v>)
//            2: partial1 = <Y, Xown>, partial2 = <Y, Y>   (omega = <t,s>/<t,t>)
//            3: partial2 = <Y, Y>                 (residual norms)
//            4: partial1 = Xown^T Y  (unconjugated; COCG sigma = p^T S p)
//            6: fused COCG: partial1 = p^T q, partial2 = q^T q (both unconjugated)   with p = Xown, q = Y
//
// Locality design (measured on cfg 3 with rocprofv3 FETCH_SIZE: a one-row-per-wave kernel
// over all 64 columns re-fetched every gathered X row ~5x from beyond L2 -- 5.2 GB per launch
// against 1.75 GB algorithmic):
//   * persistent 1-D grid of 8*S workgroups; block b belongs to XCD group (b & 7).  Each XCD
//     group owns ONE 16-column tile of the panel and one slice of the rows (8/NT slices), so
//     a gathered X line is only ever wanted by one private L2;
//   * inside a group the S workgroups sweep the slice as a moving band of S*16 consecutive
//     rows, so the X lines live in L2 for the stencil reuse distance only (band + halo,
//     ~1.5 MB at cfg 3) and every X line is fetched from HBM once;
//   * streaming operands (U, Bvec in; Y out) use non-temporal accesses so they do not evict
//     the X window or the CSR slice;
//   * a wave covers 4 rows x 16 columns; the 16 lanes of a row load 16 nonzeros' column
//     index / A / B values in one coalesced access and broadcast them with shuffles.  When the
//     coefficient pair is the same for every column (the shifted operator z B - A) the loading
//     lane forms s = cb*b + ca*a once per nonzero.
// Placement (which XCD a block lands on) only affects speed, never results.
// ------------------------------------------------------------------------------------
static inline int fh_spmm_slots(int N, int ld) {
    const int nt = ld / 16, slices = 8 / nt;
    int rows = (N + slices - 1) / slices;
    int S = (rows + 15) / 16;
    if (S > 128) S = 128;     // 8*S <= 1024 workgroups, all co-resident at 4 waves/SIMD
    if (S < 1) S = 1;
    return S;
}
int fh_spmm_grid(int N, int ld) { return 8 * fh_spmm_slots(N, ld); }
// number of partial-sum rows per node the finalize kernels must add up
int fh_spmm_partials(int N, int ld) { return (8 / (ld / 16)) * fh_spmm_slots(N, ld); }

__device__ __forceinline__ double fh_shfl16(double v, int src) { return __shfl(v, src, 16); }
__device__ __forceinline__ cplx fh_shfl16(cplx v, int src) { return cmake(__shfl(v.x, src, 16), __shfl(v.y, src, 16)); }
__device__ __forceinline__ cplxf fh_shfl16(cplxf v, int src) { return cmakef(__shfl(v.x, src, 16), __shfl(v.y, src, 16)); }
// Broadcast of lane q of every 16-lane row to the whole row as a DPP move (v_mov_b32_dpp row_newbcast:q): a VALU
// instruction, where __shfl(v, q, 16) is a ds_bpermute through the LDS pipe -- 40 of those per matrix row in the SpMM
// inner loop (8 column indices + 8 complex coefficients).  q must fold to a constant (unrolled loops).
template <int Q> __device__ __forceinline__ int fh_bc16c(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + Q, 0xF, 0xF, false); }
__device__ __forceinline__ int fh_bc16(int v, int q) {
    switch (q & 15) {
        case 0: return fh_bc16c<0>(v); case 1: return fh_bc16c<1>(v); case 2: return fh_bc16c<2>(v); case 3: return fh_bc16c<3>(v);
        case 4: return fh_bc16c<4>(v); case 5: return fh_bc16c<5>(v); case 6: return fh_bc16c<6>(v); case 7: return fh_bc16c<7>(v);
        case 8: return fh_bc16c<8>(v); case 9: return fh_bc16c<9>(v); case 10: return fh_bc16c<10>(v); case 11: return fh_bc16c<11>(v);
        case 12: return fh_bc16c<12>(v); case 13: return fh_bc16c<13>(v); case 14: return fh_bc16c<14>(v); default: return fh_bc16c<15>(v);
    }
}
__device__ __forceinline__ float fh_bc16(float v, int q) { return __int_as_float(fh_bc16(__float_as_int(v), q)); }
__device__ __forceinline__ double fh_bc16(double v, int q) {
    return __hiloint2double(fh_bc16(__double2hiint(v), q), fh_bc16(__double2loint(v), q));
}
__device__ __forceinline__ cplx fh_bc16(cplx v, int q) { return cmake(fh_bc16(v.x, q), fh_bc16(v.y, q)); }
__device__ __forceinline__ cplxf fh_bc16(cplxf v, int q) { return cmakef(fh_bc16(v.x, q), fh_bc16(v.y, q)); }
__device__ __forceinline__ double fh_vzero(double) { return 0.0; }
__device__ __forceinline__ cplx fh_vzero(cplx) { return cmake(0, 0); }
__device__ __forceinline__ cplx fh_ld_nt(const cplx* p) {
    cplx r;
    r.x = __builtin_nontemporal_load(&p->x);
    r.y = __builtin_nontemporal_load(&p->y);
    return r;
}
__device__ __forceinline__ cplxf fh_ld_nt(const cplxf* p) {
    cplxf r;
    r.x = __builtin_nontemporal_load(&p->x);
    r.y = __builtin_nontemporal_load(&p->y);
    return r;
}
__device__ __forceinline__ void fh_st_nt(cplx* p, cplx v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
}
__device__ __forceinline__ void fh_st_nt(cplxf* p, cplxf v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
}
template <typename CT> __device__ __forceinline__ CT fh_czero();
template <> __device__ __forceinline__ cplx fh_czero<cplx>() { return cmake(0, 0); }
template <> __device__ __forceinline__ cplxf fh_czero<cplxf>() { return cmakef(0.f, 0.f); }

// SCALED: the lazy start's first product (fh_spmm_args::colscale); its own instantiation -- the two extra registers of the
// column factor spill 6-14 VGPRs at this kernel's 128-VGPR budget
template <typename CT, typename VT, int LD, bool BIDENT, bool SCALED = false>
__global__ __launch_bounds__(FH_BLOCK, 4) void k_spmm(fh_spmm_args a) {
    constexpr int NT = LD / 16;          // column tiles
    constexpr int SLICES = 8 / NT;       // row slices
    __shared__ cplx red[FH_BLOCK];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int S = gridDim.x >> 3;
    const int grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = grp % NT, slice = grp / NT;
    const int slice_rows = (a.N + SLICES - 1) / SLICES;
    const int row_lo = min(a.N, slice * slice_rows);
    const int row_hi = min(a.N, row_lo + slice_rows);
    const int band = S * 16;
    const int c = ct * 16 + l16;
    const VT* __restrict__ aval = (const VT*)a.aval;
    const VT* __restrict__ bval = (const VT*)a.bval;
    const int* __restrict__ rowptr = a.rowptr;
    const int* __restrict__ colidx = a.col;
    const int prow = slice * S + slot;                   // partial-sum row of this block
    const int nprow = SLICES * S;

    for (int node = 0; node < a.nodes; ++node) {
        const size_t pbase = ((size_t)node * nprow + prow) * LD + ct * 16;
        if (a.node_active && a.node_active[node] == 0) {
            if (a.dot_mode != 0 && t < 16) {
                if (a.partial1) a.partial1[pbase + t] = cmake(0, 0);
                if (a.partial2) a.partial2[pbase + t] = cmake(0, 0);
            }
            continue;
        }
        if (a.counters && blockIdx.x == 0 && t == 0) {
            // measurement support: one matrix sweep per active node plus
            // (vector passes) x (active columns) column sweeps
            const int cols = a.node_active ? a.node_active[node] : a.m;
            const int passes = (a.dot_mode == 1 || a.Bvec) ? 3 : 2;
            atomicAdd(a.counters + 0, 1ull);
            atomicAdd(a.counters + 1, (unsigned long long)(cols * passes));
        }
        const CT* __restrict__ X = (const CT*)a.X + (size_t)node * a.x_node_stride;
        CT* __restrict__ Y = (CT*)a.Y + (size_t)node * a.y_node_stride;
        const CT* __restrict__ Bv = a.Bvec ? (const CT*)a.Bvec + (size_t)node * a.b_node_stride : nullptr;
        const CT* __restrict__ U = a.U ? (const CT*)a.U + (size_t)node * a.u_node_stride : nullptr;
        const CT ca = cvt<CT>(a.coefA[node * LD + c]);
        const CT cb = cvt<CT>(a.coefB[node * LD + c]);
        constexpr bool scaled = SCALED;                       // lazy start: the panel is colscale[node][column] * X (see k_spmm_row)
        const CT fsc = scaled ? cvt<CT>(a.colscale[node * LD + c]) : fh_czero<CT>();
        cplx d1 = cmake(0, 0), d2 = cmake(0, 0);
        // software pipeline over this group's rows: row pointers are fetched two rows ahead and
        // the first 16 nonzeros' (column, A, B) one row ahead, so a row's critical path is the
        // X gather alone instead of rowptr -> (col, val) -> gather in series
        const int i_first = row_lo + slot * 16 + wave * 4 + g;
        int k0n = 0, k1n = 0, k0nn = 0, k1nn = 0;
        if (i_first < row_hi) { k0n = rowptr[i_first]; k1n = rowptr[i_first + 1]; }
        if (i_first + band < row_hi) { k0nn = rowptr[i_first + band]; k1nn = rowptr[i_first + band + 1]; }
        // (slots past the end of a row hold the row's OWN index and zero values: the gather of such a slot re-reads the
        //  row's cached line and contributes nothing -- no select in the inner loop)
        int ncol = i_first; VT na = fh_vzero(VT()), nb = fh_vzero(VT());
        if (i_first < row_hi && k0n + l16 < k1n) {
            ncol = colidx[k0n + l16]; na = aval[k0n + l16];
            if (!BIDENT) nb = bval[k0n + l16];
        }
        // FAR GATHER, one row step ahead.  Ingest puts the nonzero with the largest column index first in its row
        // (fh_api.hip): in an ascending sweep that is the one X row of the step that no workgroup has touched yet -- it
        // comes from HBM, the other gathers from L1/L2 -- and a wave that waits for it every step keeps exactly one HBM
        // request in flight (measured: 2.9 us per 4-row step whatever the row length, 2.0-2.5 TB/s algorithmic).  Slot 0 of
        // the NEXT row is therefore gathered while this row is computed: issued after this row's own gathers, so the
        // wait for those (vmcnt counts in issue order) leaves it in flight.
        CT xfar = fh_czero<CT>();
        if (i_first < row_hi) xfar = X[(size_t)fh_bc16(ncol, 0) * LD + c];
        for (int i = i_first; i < row_hi; i += band) {
            const int k0 = k0n, k1 = k1n;
            int mycol = ncol; VT mya = na, myb = nb;
            const CT xfar_cur = xfar;
            // prefetch for the next row of this group
            k0n = k0nn; k1n = k1nn;
            if (i + 2 * band < row_hi) { k0nn = rowptr[i + 2 * band]; k1nn = rowptr[i + 2 * band + 1]; }
            ncol = i + band; na = fh_vzero(VT()); nb = fh_vzero(VT());
            if (i + band < row_hi && k0n + l16 < k1n) {
                ncol = colidx[k0n + l16]; na = aval[k0n + l16];
                if (!BIDENT) nb = bval[k0n + l16];
            }
            CT acc = fh_czero<CT>();
            const CT xown = X[(size_t)i * LD + c];
            if (BIDENT) acc = cmul(cb, xown);            // B = I contributes cb * x_i
            for (int kb = k0; kb < k1; kb += 16) {
                if (kb != k0) {                          // rows longer than 16 nonzeros: load on demand
                    const int kk = kb + l16;
                    const bool in = kk < k1;
                    mycol = in ? colidx[kk] : i;
                    mya = in ? aval[kk] : fh_vzero(VT());
                    myb = fh_vzero(VT());
                    if (!BIDENT) myb = in ? bval[kk] : fh_vzero(VT());
                }
                CT mys = fh_czero<CT>();
                if (a.uniform_coef) {                     // same (ca, cb) in every lane of the row
                    mys = vmul(mya, ca);
                    if (!BIDENT) mys = cadd(mys, vmul(myb, cb));
                }
                const int cnt = min(16, k1 - kb);
                // batches of 8 nonzeros: issue all gathers first, then the FMAs.  Slots past the
                // row end carry a zero coefficient and gather the row's own (cached) line.
#pragma unroll
                for (int q0 = 0; q0 < 16; q0 += 8) {
                    if (q0 >= cnt) break;
                    CT xs[8];
                    const bool head = (q0 == 0) && (kb == k0);          // slot 0 of the row arrived a step ago
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (q == 0 && q0 == 0) {
                            // (kb == k0 on the first trip of every lane: the branch is uniform)
                            if (head) { xs[0] = xfar_cur; continue; }
                        }
                        const int j = fh_bc16(mycol, q0 + q);
                        // the diagonal entry and the slots past the row end point at the row itself: its value is in a
                        // register already (xown) -- no second and third request for the same line (the kernel is bound
                        // by its requests: 7 per stencil row instead of 10)
                        xs[q] = xown;
                        if (j != i) xs[q] = X[(size_t)j * LD + c];
                    }
                    if (head) {
                        // next row's far gather: the youngest load of this step
                        xfar = fh_czero<CT>();
                        if (i + band < row_hi) xfar = X[(size_t)fh_bc16(ncol, 0) * LD + c];
                    }
                    if (a.uniform_coef) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) cfma(acc, fh_bc16(mys, q0 + q), xs[q]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            CT sc = vmul(fh_bc16(mya, q0 + q), ca);        // zero beyond the row end
                            if (!BIDENT) sc = cadd(sc, vmul(fh_bc16(myb, q0 + q), cb));
                            cfma(acc, sc, xs[q]);
                        }
                    }
                }
            }
            if (k1 == k0 && i + band < row_hi) xfar = X[(size_t)fh_bc16(ncol, 0) * LD + c];   // empty row: nothing issued above
            CT xo = xown;
            if (scaled) { acc = cmul(fsc, acc); xo = cmul(fsc, xown); }       // S (x diag(f)) = (S x) diag(f)
            if (Bv) acc = csub(fh_ld_nt(Bv + (size_t)i * LD + c), acc);
            fh_st_nt(Y + (size_t)i * LD + c, acc);
            const cplx accd = to_d(acc);
            if (a.dot_mode == 1) {
                d1 = cadd(d1, cmulc(to_d(fh_ld_nt(U + (size_t)i * LD + c)), accd));
            } else if (a.dot_mode == 2) {
                d1 = cadd(d1, cmulc(accd, to_d(xo)));
                d2.x += cabs2(accd);
            } else if (a.dot_mode == 3) {
                d2.x += cabs2(accd);
            } else if (a.dot_mode == 4) {
                d1 = cadd(d1, cmul(to_d(xo), accd));   // unconjugated p^T (S p), COCG
            } else if (a.dot_mode == 6) {
                d1 = cadd(d1, cmul(to_d(xo), accd));   // p^T q
                d2 = cadd(d2, cmul(accd, accd));         // q^T q (unconjugated)
            }
        }
        if (a.dot_mode != 0) {
            // per-column block reduction: 16 threads (4 waves x 4 row groups) share a column
            if (a.dot_mode == 1 || a.dot_mode == 2 || a.dot_mode == 4 || a.dot_mode == 6) {
                red[t] = d1;
                __syncthreads();
                if (t < 16) {
                    cplx s = cmake(0, 0);
                    for (int k = 0; k < 16; ++k) s = cadd(s, red[t + 16 * k]);
                    a.partial1[pbase + t] = s;
                }
                __syncthreads();
            }
            if (a.dot_mode == 2 || a.dot_mode == 3 || a.dot_mode == 6) {
                red[t] = d2;
                __syncthreads();
                if (t < 16) {
                    cplx s = cmake(0, 0);
                    for (int k = 0; k < 16; ++k) s = cadd(s, red[t + 16 * k]);
                    a.partial2[pbase + t] = s;
                }
                __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Row-per-wave SpMM for full-width panels (LD = 64, real matrix values): one WAVE takes one matrix row, lane = column.
//
// Why a second gather kernel.  k_spmm gives every 16-lane group of a wave its own row, so column indices and values live
// in VGPRs and reach the lanes of a row through 40 DPP broadcasts per row; with 64-bit address arithmetic per gather that
// is ~550 VALU cycles per 4-row step and 122-128 VGPRs (4 waves per SIMD), and the wave waits for its one HBM gather
// every step.  Here the whole wave works on ONE row, so everything about the row is wave-uniform: row bounds, column
// indices, A and B values are SCALAR loads into SGPRs, the gather base address of each nonzero is SALU arithmetic, the
// coefficient of an FMA is an SGPR operand (v_fma_f64 with the matrix value straight from the scalar file) and no lane
// ever broadcasts anything.  The products are accumulated as u = A x and v = B x (real coefficient times complex value:
// two FMAs each per nonzero) and combined once per row, y = cb v + ca u, with PER-LANE complex coefficients -- the same
// kernel serves (z B - A) X, A X, B X and the residual with one coefficient pair per column.  VALU work per row drops to
// the 32 FMAs of the products plus the dots.
// Locality: XCD group g = blockIdx & 7 owns the row slice g; its workgroups sweep the slice as a moving band of
// (workgroups per group) x 16 rows, so the X rows a band needs (band + the two stencil planes) stay in that XCD's L2 or,
// for the far planes of a wide stencil, in the Infinity Cache.  Row order inside a row: the far gather (largest column,
// first in its row: fh_api.hip ingest) is issued first so that it has the longest time to arrive.
// ------------------------------------------------------------------------------------
#ifndef FH_ROW_THREADS
#define FH_ROW_THREADS 1024
#endif
// Measured (cfg 3, us per launch at 1 / 3 / 16 active nodes; k_spmm: 33.3 / 128.4 / 582.8):
//   1024 threads, 4 waves/SIMD (121 VGPRs, no spills), 1 workgroup per CU:  36.5 / 120.6 / 510.9   <- default
//    512 threads, 6 waves/SIMD ( 79 VGPRs, no spills), 3 per CU:            34.7 / 117.1 / 575.2
//   8 waves/SIMD (64 VGPRs): 73-80 VGPRs spilled (the kernel arguments alone hold ~50 SGPRs):  50 / 137-145 / 565-630
#ifndef FH_ROW_WAVES
#define FH_ROW_WAVES 4                     // waves per SIMD the register budget is cut for
#endif
#ifndef FH_ROW_BLOCKS_MAX
#define FH_ROW_BLOCKS_MAX 1                // resident workgroups per CU the grid is sized for
#endif
static inline int fh_spmm_row_groups(int N, int blocks_per_cu) {
    (void)N;
    return 32 * blocks_per_cu;                 // workgroups per XCD group: every CU of the XCD holds blocks_per_cu of them
}

__device__ __forceinline__ double fh_uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// UNIF: the coefficient pair is the same for every column (the shifted operator z B - A): it lives in SGPRs
template <typename CT, bool BIDENT, bool UNIF>
__global__ __launch_bounds__(FH_ROW_THREADS, FH_ROW_WAVES) void k_spmm_row(fh_spmm_args a) {
    constexpr int LD = 64;
    __shared__ double red[FH_ROW_THREADS / 64][64][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // uniform: everything below it is SALU
    constexpr int WPB = FH_ROW_THREADS / 64;
    const int S = gridDim.x >> 3;
    const int grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int slice_rows = (a.N + 7) / 8;
    const int row_lo = min(a.N, grp * slice_rows);
    const int row_hi = min(a.N, row_lo + slice_rows);
    const int band = S * WPB;
    // the matrix arrays are read-only for the whole launch: constant address space, so that a load with a wave-uniform
    // address is a scalar load (s_load_dwordx8 / x16) into SGPRs instead of a vector load that every lane repeats
    typedef const double __attribute__((address_space(4))) * cdptr;
    typedef const int __attribute__((address_space(4))) * ciptr;
    const cdptr a8 = (cdptr)a.a8;
    const cdptr b8 = (cdptr)a.b8;
    const ciptr rp8 = (ciptr)a.rp8;
    const ciptr col8 = (ciptr)a.col8;
    const int prow = blockIdx.x, nprow = gridDim.x;

    for (int node = 0; node < a.nodes; ++node) {
        const size_t pbase = ((size_t)node * nprow + prow) * LD;
        if (a.node_active && a.node_active[node] == 0) {
            if (a.dot_mode != 0 && threadIdx.x < LD) {
                if (a.partial1) a.partial1[pbase + threadIdx.x] = cmake(0, 0);
                if (a.partial2) a.partial2[pbase + threadIdx.x] = cmake(0, 0);
            }
            continue;
        }
        if (a.counters && blockIdx.x == 0 && threadIdx.x == 0) {
            const int cols = a.node_active ? a.node_active[node] : a.m;
            const int passes = (a.dot_mode == 1 || a.Bvec) ? 3 : 2;
            atomicAdd(a.counters + 0, 1ull);
            atomicAdd(a.counters + 1, (unsigned long long)(cols * passes));
        }
        const CT* __restrict__ X = (const CT*)a.X + (size_t)node * a.x_node_stride;
        CT* __restrict__ Y = (CT*)a.Y + (size_t)node * a.y_node_stride;
        const CT* __restrict__ Bv = a.Bvec ? (const CT*)a.Bvec + (size_t)node * a.b_node_stride : nullptr;
        const CT* __restrict__ U = a.U ? (const CT*)a.U + (size_t)node * a.u_node_stride : nullptr;
        cplx ca = a.coefA[node * LD + lane];
        cplx cb = a.coefB[node * LD + lane];
        if (UNIF) { ca = cmake(fh_uniform(ca.x), fh_uniform(ca.y)); cb = cmake(fh_uniform(cb.x), fh_uniform(cb.y)); }
        const bool scaled = a.colscale != nullptr;              // lazy start: the panel is colscale[node][column] * X
        const cplx fs = scaled ? a.colscale[node * LD + lane] : cmake(1, 0);
        double d1x = 0.0, d1y = 0.0, d2x = 0.0, d2y = 0.0;
        for (int i = row_lo + slot * WPB + wave; i < row_hi; i += band) {
            const int c0 = rp8[i], c1 = rp8[i + 1];
            double ux = 0.0, uy = 0.0, vx = 0.0, vy = 0.0;
            // the row's own X value once; the diagonal entry and the padding slots of a chunk (column == own row) take
            // it from the register instead of gathering the same 1-KB line again: 7 requests per row of the 7-point
            // stencil instead of 10 -- the kernel is bound by the rate of exactly these requests.
            // The test is wave-uniform (scalar compare and branch).
            const CT xown = (X + (size_t)i * LD)[lane];
            for (int ch = c0; ch < c1; ++ch) {
                const ciptr cc = col8 + (size_t)ch * 8;
                const cdptr aa = a8 + (size_t)ch * 8;
                CT xs[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = cc[q];
                    if (j != i) {
                        const CT* __restrict__ rowp = X + (size_t)j * LD;        // uniform base: SALU
                        xs[q] = rowp[lane];
                    } else {
                        xs[q] = xown;
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const cplx x = to_d(xs[q]);
                    const double av = aa[q];
                    ux += av * x.x; uy += av * x.y;
                    if (!BIDENT) {
                        const double bv = (b8 + (size_t)ch * 8)[q];
                        vx += bv * x.x; vy += bv * x.y;
                    }
                }
            }
            cplx u = cmake(ux, uy), v = cmake(vx, vy);
            if (BIDENT) v = to_d(xown);
            cplx accd = cadd(cmul(cb, v), cmul(ca, u));
            cplx xo = to_d(xown);
            if (scaled) { accd = cmul(fs, accd); xo = cmul(fs, xo); }       // S (x diag(f)) = (S x) diag(f)
            if (Bv) accd = csub(to_d(fh_ld_nt((Bv + (size_t)i * LD) + lane)), accd);
            const CT acc = cvt<CT>(accd);
            fh_st_nt((Y + (size_t)i * LD) + lane, acc);
            accd = to_d(acc);
            if (a.dot_mode == 1) {
                const cplx t1 = cmulc(to_d(fh_ld_nt((U + (size_t)i * LD) + lane)), accd);
                d1x += t1.x; d1y += t1.y;
            } else if (a.dot_mode == 2) {
                const cplx t1 = cmulc(accd, xo);
                d1x += t1.x; d1y += t1.y; d2x += cabs2(accd);
            } else if (a.dot_mode == 3) {
                d2x += cabs2(accd);
            } else if (a.dot_mode == 4) {
                const cplx t1 = cmul(xo, accd);
                d1x += t1.x; d1y += t1.y;
            } else if (a.dot_mode == 6) {
                const cplx t1 = cmul(xo, accd), t2 = cmul(accd, accd);
                d1x += t1.x; d1y += t1.y; d2x += t2.x; d2y += t2.y;
            }
        }
        if (a.dot_mode != 0) {
            // lane = column: the 16 waves of the workgroup hold 16 partial sums per column, added in wave order
            red[wave][lane][0] = d1x; red[wave][lane][1] = d1y; red[wave][lane][2] = d2x; red[wave][lane][3] = d2y;
            __syncthreads();
            if (threadIdx.x < LD) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int w = 0; w < WPB; ++w) { s0 += red[w][threadIdx.x][0]; s1 += red[w][threadIdx.x][1]; s2 += red[w][threadIdx.x][2]; s3 += red[w][threadIdx.x][3]; }
                if (a.partial1) a.partial1[pbase + threadIdx.x] = cmake(s0, s1);
                if (a.partial2) a.partial2[pbase + threadIdx.x] = cmake(s2, s3);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------
// LDS-window SpMM for a matrix that ingest renumbered into row blocks (fh_api.hip: fh_block_partition; complex128 panels).
// OPT-IN (FH_REORDER=1|2): measured SLOWER than k_spmm on cfg 3 -- 57 us against 33 us per node -- see below.
//
// k_spmm gathers every X row a matrix row touches from L1/L2: 8 gathers of 256 B per row and 16-column tile.  Here a
// workgroup takes one row block (<= 128 rows) x one 16-column tile and first pulls every X row the block needs -- its
// own rows and the distinct outside rows its nonzeros touch, listed at ingest (129 per block on cfg 3 after the
// bisection renumbering) -- into LDS with LDS-DMA row gathers (global_load_lds_dwordx4: per-lane source address, 1 KiB =
// four 256-B rows per wave-instruction, no VGPRs, all ~260 rows of a block in flight at once), then computes from LDS
// only: the column index of a nonzero is its LDS slot (uint16).  The matrix rows of a lane group are fetched once per
// row block into registers and serve every node.  Outside rows beyond the FH_SPMM_EXT kept per block carry slot 0xFFFF
// and are gathered from global memory.  Same arithmetic order per row as k_spmm, one partial-sum row per workgroup.
//
// What the measurement says (one node, 64 columns, tools/mb_spmm.py; phases switched off one at a time): staging costs
// 7 us of the 57, the per-block set-up 15, the LDS-fed row loop 35 -- the row loop is VALU-bound (about 210 instructions
// per 4-row step: coefficient and slot broadcasts, fp64 complex FMAs, the register-array selects) at the two waves per
// SIMD that 72 KiB of LDS per workgroup leave, where k_spmm hides the same arithmetic under its gathers at four.  The
// window removes L2 requests (8 -> 2 per row), and requests were never what bounded the gather kernel: it runs at the
// same 33-35 us per node in lexicographic, brick and bisection order alike.
// ------------------------------------------------------------------------------------
#define FH_LDS_SLOTS (FH_SPMM_R + FH_SPMM_EXT)
#define FH_LDS_RPG (FH_SPMM_R / 16)          // rows per 16-lane group and row block (16 groups per workgroup)
// element s of an 8-entry register array without dynamic indexing (a select chain keeps the array in VGPRs; the row loop is
// NOT unrolled: eight unrolled rows needed 256 VGPRs and spilled)
#define FH_SEL8(arr, s) ((s) == 0 ? arr[0] : (s) == 1 ? arr[1] : (s) == 2 ? arr[2] : (s) == 3 ? arr[3] : (s) == 4 ? arr[4] : (s) == 5 ? arr[5] : (s) == 6 ? arr[6] : arr[7])
static_assert(FH_LDS_RPG == 8, "FH_SEL8 assumes eight rows per lane group");
template <typename VT, int LD, bool BIDENT>
__global__ __launch_bounds__(FH_BLOCK, 2) void k_spmm_lds(fh_spmm_args a) {
    typedef cplx CT;
    constexpr int NT = LD / 16;
    constexpr int SLICES = 8 / NT;
    extern __shared__ cplx lds_x[];                       // FH_LDS_SLOTS rows x 16 columns
    __shared__ cplx red[FH_BLOCK];
    __shared__ int s_list[64];
    __shared__ int s_cnt;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int S = gridDim.x >> 3;
    const int grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = grp % NT, slice = grp / NT;
    const int per = (a.nblk_rows + SLICES - 1) / SLICES;   // row blocks per slice
    const int b_lo = min(a.nblk_rows, slice * per), b_hi = min(a.nblk_rows, b_lo + per);
    const int c = ct * 16 + l16;
    const VT* __restrict__ aval = (const VT*)a.aval;
    const VT* __restrict__ bval = (const VT*)a.bval;
    const int* __restrict__ rowptr = a.rowptr;
    const int* __restrict__ colidx = a.col;
    const unsigned short* __restrict__ lcol = a.lcol;
    const int prow = slice * S + slot;
    const int nprow = SLICES * S;

    // active nodes of this launch; inactive ones only get their partial sums cleared
    if (t == 0) {
        int n = 0;
        for (int node = 0; node < a.nodes && n < 64; ++node)
            if (!a.node_active || a.node_active[node] != 0) s_list[n++] = node;
        s_cnt = n;
    }
    if (a.dot_mode != 0 && t < 16) {
        // cleared for EVERY node: the row blocks below add their contributions one after the other
        for (int node = 0; node < a.nodes; ++node) {
            const size_t pb = ((size_t)node * nprow + prow) * LD + ct * 16;
            if (a.partial1) a.partial1[pb + t] = cmake(0, 0);
            if (a.partial2) a.partial2[pb + t] = cmake(0, 0);
        }
    }
    __syncthreads();
    const int cnt_nodes = s_cnt;
    if (a.counters && blockIdx.x == 0 && t == 0) {
        unsigned long long cols = 0;
        for (int q = 0; q < cnt_nodes; ++q) cols += (unsigned long long)(a.node_active ? a.node_active[s_list[q]] : a.m);
        atomicAdd(a.counters + 0, (unsigned long long)cnt_nodes);
        atomicAdd(a.counters + 1, cols * (unsigned long long)((a.dot_mode == 1 || a.Bvec) ? 3 : 2));
    }

    for (int blk = b_lo + slot; blk < b_hi; blk += S) {
        const int r0 = a.blk_start[blk], nrow = a.blk_start[blk + 1] - r0;
        const int e0 = a.ext_ptr[blk], next = a.ext_ptr[blk + 1] - e0;
        const int my_hi = r0 + nrow;
        // ---- once per row block, shared by all nodes: the matrix rows of this lane group in registers (row bounds, then
        //      the first 16 nonzeros' LDS slot / A / B -- two dependent trips instead of two per row), and the outside-row
        //      indices of this wave's staging instructions
        int k0r[FH_LDS_RPG], k1r[FH_LDS_RPG], lcr[FH_LDS_RPG];
        VT ar[FH_LDS_RPG], br[FH_LDS_RPG];
#pragma unroll
        for (int s = 0; s < FH_LDS_RPG; ++s) {
            const int i = r0 + s * 16 + wave * 4 + g;
            k0r[s] = i < my_hi ? rowptr[i] : 0;
            k1r[s] = i < my_hi ? rowptr[i + 1] : 0;
        }
#pragma unroll
        for (int s = 0; s < FH_LDS_RPG; ++s) {
            const bool in = k0r[s] + l16 < k1r[s];
            lcr[s] = in ? (int)lcol[k0r[s] + l16] : 0;
            ar[s] = in ? aval[k0r[s] + l16] : fh_vzero(VT());
            br[s] = fh_vzero(VT());
            if (!BIDENT) br[s] = in ? bval[k0r[s] + l16] : fh_vzero(VT());
        }
        constexpr int NQ = (FH_SPMM_EXT / 4 + FH_BLOCK / 64 - 1) / (FH_BLOCK / 64);
        int egrow[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int sl = 4 * (wave + j * (FH_BLOCK / 64)) + g;
            egrow[j] = sl < next ? a.ext_idx[e0 + sl] : r0;
        }
        for (int nq = 0; nq < cnt_nodes; ++nq) {
            const int node = __builtin_amdgcn_readfirstlane(s_list[nq]);
            const CT* __restrict__ X = (const CT*)a.X + (size_t)node * a.x_node_stride;
            CT* __restrict__ Y = (CT*)a.Y + (size_t)node * a.y_node_stride;
            const CT* __restrict__ Bv = a.Bvec ? (const CT*)a.Bvec + (size_t)node * a.b_node_stride : nullptr;
            const CT* __restrict__ U = a.U ? (const CT*)a.U + (size_t)node * a.u_node_stride : nullptr;
            const CT ca = a.coefA[node * LD + c];
            const CT cb = a.coefB[node * LD + c];
            __syncthreads();                                  // the previous readers are done with the window
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): every index / metadata load has landed, so the
                                                              // compiler has no reason to wait between the DMA instructions
            // ---- stage: own rows into slots [0, nrow), outside rows into [R, R + next).  Every lane issues (a lane
            //      without a row re-reads row r0 into a slot nobody looks at): the LDS destination of the wave-instruction
            //      is lane 0's address + 16 * lane, so no lane may drop out.
            for (int q = wave; q * 4 < nrow; q += FH_BLOCK / 64) {
                const int sl = 4 * q + g;
                const int grow = sl < nrow ? r0 + sl : r0;
                __builtin_amdgcn_global_load_lds((const void*)(X + (size_t)grow * LD + c),
                                                 (void __attribute__((address_space(3)))*)(lds_x + (size_t)sl * 16 + l16), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int q = wave + j * (FH_BLOCK / 64);
                if (q * 4 < next) {
                    const int sl = 4 * q + g;
                    __builtin_amdgcn_global_load_lds((const void*)(X + (size_t)egrow[j] * LD + c),
                                                     (void __attribute__((address_space(3)))*)(lds_x + (size_t)(FH_SPMM_R + sl) * 16 + l16), 16, 0, 0);
                }
            }
            __syncthreads();                                  // drains vmcnt(0): the window is complete
            cplx d1 = cmake(0, 0), d2 = cmake(0, 0);
#pragma nounroll
            for (int s = 0; s < FH_LDS_RPG; ++s) {
                const int i = r0 + s * 16 + wave * 4 + g;
                if (i < my_hi) {
                    const int k0 = FH_SEL8(k0r, s), k1 = FH_SEL8(k1r, s);
                    int mylc = FH_SEL8(lcr, s); VT mya = FH_SEL8(ar, s), myb = FH_SEL8(br, s);
                    CT acc = cmake(0, 0);
                    const CT xown = lds_x[(size_t)(i - r0) * 16 + l16];
                    if (BIDENT) acc = cmul(cb, xown);
                    for (int kb = k0; kb < k1; kb += 16) {
                        if (kb != k0) {                       // rows longer than 16 nonzeros: load on demand
                            const int kk = kb + l16;
                            const bool in = kk < k1;
                            mylc = in ? (int)lcol[kk] : 0;
                            mya = in ? aval[kk] : fh_vzero(VT());
                            myb = fh_vzero(VT());
                            if (!BIDENT) myb = in ? bval[kk] : fh_vzero(VT());
                        }
                        CT mys = cmake(0, 0);
                        if (a.uniform_coef) {
                            mys = vmul(mya, ca);
                            if (!BIDENT) mys = cadd(mys, vmul(myb, cb));
                        }
                        const int cnt = min(16, k1 - kb);
#pragma unroll
                        for (int q0 = 0; q0 < 16; q0 += 8) {
                            if (q0 >= cnt) break;
                            CT xs[8];
                            int lcq[8];
                            bool ovf = false;
                            // all LDS reads of the batch issued without a branch in between (a per-nonzero branch on the
                            // overflow marker made every read wait for the previous one: 2.3 us per 4-row step)
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                lcq[q] = (q0 + q < cnt) ? fh_bc16(mylc, q0 + q) : (i - r0);
                                ovf |= lcq[q] == 0xFFFF;
                                xs[q] = lds_x[(size_t)(lcq[q] == 0xFFFF ? (i - r0) : lcq[q]) * 16 + l16];
                            }
                            if (__any(ovf)) {                 // rare: outside rows beyond the kept list come from global memory
#pragma unroll
                                for (int q = 0; q < 8; ++q)
                                    if (lcq[q] == 0xFFFF) xs[q] = X[(size_t)colidx[kb + q0 + q] * LD + c];
                            }
                            if (a.uniform_coef) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) cfma(acc, fh_bc16(mys, q0 + q), xs[q]);
                            } else {
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    CT sc = vmul(fh_bc16(mya, q0 + q), ca);
                                    if (!BIDENT) sc = cadd(sc, vmul(fh_bc16(myb, q0 + q), cb));
                                    cfma(acc, sc, xs[q]);
                                }
                            }
                        }
                    }
                    if (Bv) acc = csub(fh_ld_nt(Bv + (size_t)i * LD + c), acc);
                    fh_st_nt(Y + (size_t)i * LD + c, acc);
                    if (a.dot_mode == 1) {
                        d1 = cadd(d1, cmulc(fh_ld_nt(U + (size_t)i * LD + c), acc));
                    } else if (a.dot_mode == 2) {
                        d1 = cadd(d1, cmulc(acc, xown));
                        d2.x += cabs2(acc);
                    } else if (a.dot_mode == 3) {
                        d2.x += cabs2(acc);
                    } else if (a.dot_mode == 4) {
                        d1 = cadd(d1, cmul(xown, acc));
                    }
                }
            }
            if (a.dot_mode != 0) {
                // this row block's share of the node's partial sums, added in block order (one owner: deterministic)
                const size_t pbase = ((size_t)node * nprow + prow) * LD + ct * 16;
                if (a.dot_mode == 1 || a.dot_mode == 2 || a.dot_mode == 4) {
                    red[t] = d1;
                    __syncthreads();
                    if (t < 16) {
                        cplx sacc = a.partial1[pbase + t];
                        for (int k = 0; k < 16; ++k) sacc = cadd(sacc, red[t + 16 * k]);
                        a.partial1[pbase + t] = sacc;
                    }
                    __syncthreads();
                }
                if (a.dot_mode == 2 || a.dot_mode == 3) {
                    red[t] = d2;
                    __syncthreads();
                    if (t < 16) {
                        cplx sacc = a.partial2[pbase + t];
                        for (int k = 0; k < 16; ++k) sacc = cadd(sacc, red[t + 16 * k]);
                        a.partial2[pbase + t] = sacc;
                    }
                    __syncthreads();
                }
            }
        }
    }
}

// grid of the LDS-window kernel: 8 XCD groups x S slots, two workgroups per CU resident
int fh_spmm_lds_slots(int nblk_rows, int ld) {
    const int nt = ld / 16, slices = 8 / nt;
    int per = (nblk_rows + slices - 1) / slices;
    int S = per < 64 ? per : 64;
    return S < 1 ? 1 : S;
}

template <typename VT, int LD>
static void launch_spmm_lds(const fh_spmm_args& a, bool bident, hipStream_t st) {
    const size_t dyn = (size_t)FH_LDS_SLOTS * 16 * sizeof(cplx);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)k_spmm_lds<VT, LD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipFuncSetAttribute((const void*)k_spmm_lds<VT, LD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        attr_done = true;
    }
    dim3 grid(8 * fh_spmm_lds_slots(a.nblk_rows, LD)), block(FH_BLOCK);
    if (bident) hipLaunchKernelGGL((k_spmm_lds<VT, LD, true>), grid, block, dyn, st, a);
    else hipLaunchKernelGGL((k_spmm_lds<VT, LD, false>), grid, block, dyn, st, a);
}

template <typename CT, typename VT, int LD>
static void launch_spmm_ld(const fh_spmm_args& a, bool bident, int nblk, hipStream_t st) {
    dim3 grid(nblk), block(FH_BLOCK);
    if constexpr (sizeof(CT) == sizeof(cplx) && sizeof(VT) == sizeof(double)) {
        if (a.colscale) {                  // lazy start: COCG, so real matrices, and complex128 panels only
            if (bident) hipLaunchKernelGGL((k_spmm<CT, VT, LD, true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_spmm<CT, VT, LD, false, true>), grid, block, 0, st, a);
            return;
        }
    }
    if (bident)
        hipLaunchKernelGGL((k_spmm<CT, VT, LD, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((k_spmm<CT, VT, LD, false>), grid, block, 0, st, a);
}
template <typename CT>
static void launch_spmm_ct(const fh_spmm_args& a, int ld, bool is_complex, bool bident, int nblk, hipStream_t st) {
    if (is_complex) {
        if (ld == 16) launch_spmm_ld<CT, cplx, 16>(a, bident, nblk, st);
        else if (ld == 32) launch_spmm_ld<CT, cplx, 32>(a, bident, nblk, st);
        else launch_spmm_ld<CT, cplx, 64>(a, bident, nblk, st);
    } else {
        if (ld == 16) launch_spmm_ld<CT, double, 16>(a, bident, nblk, st);
        else if (ld == 32) launch_spmm_ld<CT, double, 32>(a, bident, nblk, st);
        else launch_spmm_ld<CT, double, 64>(a, bident, nblk, st);
    }
}
// workgroups of the row-per-wave kernel: 8 XCD groups x (32 CUs x resident workgroups per CU), capped so that a band
// step never exceeds the rows of a slice
int fh_spmm_row_grid(int N) {
    static int per_cu = 0;
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spmm_row<cplx, false, true>, FH_ROW_THREADS, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n > FH_ROW_BLOCKS_MAX ? FH_ROW_BLOCKS_MAX : n;
    }
    int S = fh_spmm_row_groups(N, per_cu);
    const int slice_rows = (N + 7) / 8, wpb = FH_ROW_THREADS / 64;
    const int need = (slice_rows + wpb - 1) / wpb;           // workgroups that cover a slice in one band step
    if (S > need) S = need < 1 ? 1 : need;
    return 8 * S;
}

void fh_launch_spmm(const fh_spmm_args& a, int ld, bool is_complex, bool bident, int nblk, hipStream_t st) {
    if (a.use_row_kernel) {                // LD = 64, real matrix values: one wave per row (k_spmm_row)
        dim3 grid(fh_spmm_row_grid(a.N)), block(FH_ROW_THREADS);
#define FH_ROW_LAUNCH(CT_, BI_) \
        do { if (a.uniform_coef) hipLaunchKernelGGL((k_spmm_row<CT_, BI_, true>), grid, block, 0, st, a); \
             else hipLaunchKernelGGL((k_spmm_row<CT_, BI_, false>), grid, block, 0, st, a); } while (0)
        if (a.prec == 32) { if (bident) FH_ROW_LAUNCH(cplxf, true); else FH_ROW_LAUNCH(cplxf, false); }
        else { if (bident) FH_ROW_LAUNCH(cplx, true); else FH_ROW_LAUNCH(cplx, false); }
#undef FH_ROW_LAUNCH
        return;
    }
    if (a.prec == 64 && a.lcol) {          // renumbered matrix, complex128 panels: the LDS-window kernel
        if (is_complex) {
            if (ld == 16) launch_spmm_lds<cplx, 16>(a, bident, st);
            else if (ld == 32) launch_spmm_lds<cplx, 32>(a, bident, st);
            else launch_spmm_lds<cplx, 64>(a, bident, st);
        } else {
            if (ld == 16) launch_spmm_lds<double, 16>(a, bident, st);
            else if (ld == 32) launch_spmm_lds<double, 32>(a, bident, st);
            else launch_spmm_lds<double, 64>(a, bident, st);
        }
        return;
    }
    if (a.prec == 32) launch_spmm_ct<cplxf>(a, ld, is_complex, bident, nblk, st);
    else launch_spmm_ct<cplx>(a, ld, is_complex, bident, nblk, st);
}

// ------------------------------------------------------------------------------------
// Krylov vector kernels.  Flat element index over the N*LD panel, grid-stride with a stride
// that is a multiple of LD so a thread always sees the same column.
// ------------------------------------------------------------------------------------
template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_init_guess(fh_vec_args a) {
    // X0[i,c] = Q[i,c] / (z_node - lambda_c)   (lambda == nullptr -> zero guess); always fp64
    const int node = blockIdx.y;
    const size_t total = (size_t)a.N * LD;
    const int c = threadIdx.x % LD;
    cplx* X = (cplx*)a.X + (size_t)node * a.node_stride;
    cplx f = cmake(0, 0);
    if (a.lambda) {
        cplx z = a.znode[node];
        f = cdiv(cmake(1, 0), cmake(z.x - a.lambda[c], z.y));
    }
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK)
        X[e] = a.lambda ? cmul(a.Q[e], f) : cmake(0, 0);
}

template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_copy_r(fh_vec_args a) {
    // Rhat = R, P = R
    const int node = blockIdx.y;
    const size_t total = (size_t)a.N * LD;
    const CT* R = (const CT*)a.R + (size_t)node * a.node_stride;
    CT* Rh = (CT*)a.Rhat + (size_t)node * a.node_stride;
    CT* P = (CT*)a.P + (size_t)node * a.node_stride;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        CT r = R[e];
        Rh[e] = r;
        P[e] = r;
    }
}

template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_p_update(fh_vec_args a) {
    // P = R + beta (P - omega V)
    const int node = blockIdx.y;
    if (a.s.node_active[node] == 0) return;
    const size_t total = (size_t)a.N * LD;
    const int c = threadIdx.x % LD;
    if (!a.s.active[node * LD + c]) return;
    const CT beta = cvt<CT>(a.s.beta[node * LD + c]), omega = cvt<CT>(a.s.omega[node * LD + c]);
    const CT* R = (const CT*)a.R + (size_t)node * a.node_stride;
    const CT* V = (const CT*)a.V + (size_t)node * a.node_stride;
    CT* P = (CT*)a.P + (size_t)node * a.node_stride;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        CT t = csub(P[e], cmul(omega, V[e]));
        P[e] = cadd(R[e], cmul(beta, t));
    }
}

template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_s_update(fh_vec_args a) {
    // S = R - alpha V
    const int node = blockIdx.y;
    if (a.s.node_active[node] == 0) return;
    const size_t total = (size_t)a.N * LD;
    const int c = threadIdx.x % LD;
    if (!a.s.active[node * LD + c]) return;
    const CT alpha = cvt<CT>(a.s.alpha[node * LD + c]);
    const CT* R = (const CT*)a.R + (size_t)node * a.node_stride;
    const CT* V = (const CT*)a.V + (size_t)node * a.node_stride;
    CT* S = (CT*)a.S + (size_t)node * a.node_stride;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK)
        S[e] = csub(R[e], cmul(alpha, V[e]));
}

template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_xr_update(fh_vec_args a) {
    // X += alpha P + omega S ; R = S - omega T ; partial1 = <Rhat, R>, partial2 = <R, R>
    const int node = blockIdx.y;
    const int c = threadIdx.x % LD;
    if (a.counters && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.counters + 2, (unsigned long long)a.s.node_active[node]);
    const size_t o = ((size_t)node * gridDim.x + blockIdx.x) * LD;
    cplx d1 = cmake(0, 0), d2 = cmake(0, 0);
    const bool on = a.s.node_active[node] != 0 && a.s.active[node * LD + c];
    if (on) {
        const size_t total = (size_t)a.N * LD;
        const CT alpha = cvt<CT>(a.s.alpha[node * LD + c]), omega = cvt<CT>(a.s.omega[node * LD + c]);
        const CT* P = (const CT*)a.P + (size_t)node * a.node_stride;
        const CT* S = (const CT*)a.S + (size_t)node * a.node_stride;
        const CT* T = (const CT*)a.T + (size_t)node * a.node_stride;
        const CT* Rh = (const CT*)a.Rhat + (size_t)node * a.node_stride;
        CT* X = (CT*)a.X + (size_t)node * a.node_stride;
        CT* R = (CT*)a.R + (size_t)node * a.node_stride;
        for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
            CT s = S[e];
            CT x = X[e];
            cfma(x, alpha, P[e]);
            cfma(x, omega, s);
            X[e] = x;
            CT r = csub(s, cmul(omega, T[e]));
            R[e] = r;
            const cplx rd = to_d(r);
            d1 = cadd(d1, cmulc(to_d(Rh[e]), rd));
            d2.x += cabs2(rd);
        }
    }
    __shared__ cplx red[FH_BLOCK];
    fh_block_reduce_cols<LD>(d1, red, a.partial1 + o);
    fh_block_reduce_cols<LD>(d2, red, a.partial2 + o);
}

// ---- mixed precision: scaled narrowing of the fp64 residual, widening update of X ----------
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_narrow_scaled(const cplx* __restrict__ src, size_t src_stride,
                                                             cplxf* __restrict__ dst, size_t dst_stride,
                                                             const double* __restrict__ r0norm, size_t total) {
    // dst[node][e] = src[node][e] / ||r0_c||   (zero columns stay zero)
    const int node = blockIdx.y;
    const double n = r0norm[node * LD + threadIdx.x % LD];
    const double inv = n > 0.0 ? 1.0 / n : 0.0;
    const cplx* s = src + (size_t)node * src_stride;
    cplxf* d = dst + (size_t)node * dst_stride;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        cplx v = s[e];
        d[e] = cmakef((float)(v.x * inv), (float)(v.y * inv));
    }
}
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_widen_axpy(cplx* __restrict__ X, size_t x_stride,
                                                          const cplxf* __restrict__ D, size_t d_stride,
                                                          const double* __restrict__ r0norm, size_t total) {
    // X[node][e] += ||r0_c|| * D[node][e]
    const int node = blockIdx.y;
    const double n = r0norm[node * LD + threadIdx.x % LD];
    cplx* x = X + (size_t)node * x_stride;
    const cplxf* d = D + (size_t)node * d_stride;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        cplx v = x[e];
        cplxf w = d[e];
        x[e] = cmake(v.x + n * (double)w.x, v.y + n * (double)w.y);
    }
}

// ---- finalize kernels: one block per node, reduce the per-block partials (always fp64) -------
template <int LD>
__device__ __forceinline__ cplx fh_sum_partials(const cplx* partial, int node, int nblk, cplx* red) {
    const int t = threadIdx.x;
    const int c = t % LD, g = t / LD;
    constexpr int G = FH_FIN_BLOCK / LD;
    cplx s0 = cmake(0, 0), s1 = cmake(0, 0);
    const cplx* p = partial + (size_t)node * nblk * LD + c;
    int b = g;
    // The partial rows were written by other XCDs: every load is a trip to HBM / the Infinity Cache (1-2 us).  Eight
    // independent loads in flight per thread, summed in a fixed order (deterministic): the kernel is a latency chain,
    // its length is the number of such batches (2 for the 256 rows of a 16-node sweep), not the bytes.
    for (; b + 7 * G < nblk; b += 8 * G) {
        cplx v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(b + q * G) * LD];
        s0 = cadd(s0, cadd(cadd(v[0], v[1]), cadd(v[2], v[3])));
        s1 = cadd(s1, cadd(cadd(v[4], v[5]), cadd(v[6], v[7])));
    }
    for (; b < nblk; b += G) s0 = cadd(s0, p[(size_t)b * LD]);
    red[t] = cadd(s0, s1);
    __syncthreads();
    cplx tot = cmake(0, 0);
    if (t < LD) {
        tot = red[t];
        for (int k = 1; k < G; ++k) tot = cadd(tot, red[t + k * LD]);
    }
    __syncthreads();
    return tot;  // valid for t < LD
}

// two partial arrays in one pass (one barrier pair instead of two, loads of both arrays overlap)
template <int LD>
__device__ __forceinline__ void fh_sum_partials2(const cplx* partial1, const cplx* partial2, int node, int nblk, cplx* red,
                                                 cplx& tot1, cplx& tot2) {
    const int t = threadIdx.x;
    const int c = t % LD, g = t / LD;
    constexpr int G = FH_FIN_BLOCK / LD;
    cplx a0 = cmake(0, 0), a1 = cmake(0, 0), b0 = cmake(0, 0), b1 = cmake(0, 0);
    const cplx* p = partial1 + (size_t)node * nblk * LD + c;
    const cplx* q = partial2 + (size_t)node * nblk * LD + c;
    int b = g;
    for (; b + 3 * G < nblk; b += 4 * G) {       // 4 + 4 independent loads in flight (see fh_sum_partials)
        cplx u[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { u[k] = p[(size_t)(b + k * G) * LD]; w[k] = q[(size_t)(b + k * G) * LD]; }
        a0 = cadd(a0, cadd(u[0], u[1])); a1 = cadd(a1, cadd(u[2], u[3]));
        b0 = cadd(b0, cadd(w[0], w[1])); b1 = cadd(b1, cadd(w[2], w[3]));
    }
    for (; b < nblk; b += G) { a0 = cadd(a0, p[(size_t)b * LD]); b0 = cadd(b0, q[(size_t)b * LD]); }
    red[t] = cadd(a0, a1);
    red[FH_FIN_BLOCK + t] = cadd(b0, b1);
    __syncthreads();
    tot1 = cmake(0, 0); tot2 = cmake(0, 0);
    if (t < LD) {
        tot1 = red[t]; tot2 = red[FH_FIN_BLOCK + t];
        for (int k = 1; k < G; ++k) { tot1 = cadd(tot1, red[t + k * LD]); tot2 = cadd(tot2, red[FH_FIN_BLOCK + t + k * LD]); }
    }
    __syncthreads();
}

__device__ __forceinline__ bool fh_finite(cplx a) { return isfinite(a.x) && isfinite(a.y); }

// mode 0: BiCGStab (rho = ||r||^2 from partial2); mode 1: COCG (rho = r^T r from partial1)
template <int LD>
__global__ __launch_bounds__(FH_FIN_BLOCK) void k_fin_init(fh_fin_args a) {
    __shared__ cplx red[FH_FIN_BLOCK];
    __shared__ int cnt;
    const int node = blockIdx.x, t = threadIdx.x;
    if (t == 0) cnt = 0;
    cplx rho1 = cmake(0, 0);
    if (a.mode == 1) rho1 = fh_sum_partials<LD>(a.partial1, node, a.nblk, red);
    cplx rr = fh_sum_partials<LD>(a.partial2, node, a.nblk, red);
    if (t < LD) {
        const int i = node * LD + t;
        double rn = sqrt(rr.x);
        a.s.r0norm[i] = rn;
        a.s.rnorm[i] = rn;
        double target = a.rtol * rn + a.atol * (a.atol_scale ? a.atol_scale[i] : 1.0);
        a.s.target[i] = target;
        a.s.rho[i] = a.mode == 1 ? rho1 : cmake(rr.x, 0);
        a.s.alpha[i] = cmake(1, 0);
        a.s.omega[i] = cmake(1, 0);
        a.s.beta[i] = cmake(0, 0);
        a.s.iters[i] = 0;
        int act = (t < a.m) && (rn > target) && isfinite(rn) && (!a.col_mask || a.col_mask[t]);
        a.s.active[i] = act;
        a.s.status[i] = (t < a.m && !isfinite(rn)) ? 8 : 0;
        if (act) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (t == 0) a.s.node_active[node] = cnt;
}

template <int LD>
__global__ __launch_bounds__(FH_FIN_BLOCK) void k_fin_alpha(fh_fin_args a) {
    // alpha = rho / sigma   (sigma = <rhat, v> for BiCGStab, p^T S p for COCG)
    __shared__ cplx red[FH_FIN_BLOCK];
    const int node = blockIdx.x, t = threadIdx.x;
    if (a.s.node_active[node] == 0) {
        if (a.s.accum) {
            if (t < LD) a.s.accum[node * LD + t] = 0;
            if (t == 0) a.s.node_accum[node] = 0;
        }
        return;
    }
    cplx sigma = fh_sum_partials<LD>(a.partial1, node, a.nblk, red);
    if (t < LD) {
        const int i = node * LD + t;
        int stepped = 0;
        if (a.s.active[i]) {
            cplx al = cdiv(a.s.rho[i], sigma);
            if (cabs2(sigma) == 0.0 || !fh_finite(al)) {
                a.s.active[i] = 0;
                a.s.status[i] = 8;   // breakdown
                al = cmake(0, 0);
            } else {
                stepped = 1;
            }
            a.s.alpha[i] = al;
        }
        if (a.s.accum) a.s.accum[i] = stepped;
    }
    if (t == 0 && a.s.accum) a.s.node_accum[node] = 1;
}

template <int LD>
__global__ __launch_bounds__(FH_FIN_BLOCK) void k_fin_omega(fh_fin_args a) {
    // omega = <t,s> / <t,t>
    __shared__ cplx red[FH_FIN_BLOCK];
    const int node = blockIdx.x, t = threadIdx.x;
    if (a.s.node_active[node] == 0) return;
    cplx ts = fh_sum_partials<LD>(a.partial1, node, a.nblk, red);
    cplx tt = fh_sum_partials<LD>(a.partial2, node, a.nblk, red);
    if (t < LD) {
        const int i = node * LD + t;
        if (a.s.active[i]) {
            cplx om = cmake(ts.x / tt.x, ts.y / tt.x);
            if (tt.x == 0.0 || !fh_finite(om)) om = cmake(0, 0);  // s == 0: x += alpha p is exact
            a.s.omega[i] = om;
        }
    }
}

template <int LD>
__global__ __launch_bounds__(FH_FIN_BLOCK) void k_fin_rho(fh_fin_args a) {
    // mode 0: rho_new = <rhat, r>, beta = (rho_new/rho)(alpha/omega)   (BiCGStab)
    // mode 1: rho_new = r^T r,     beta = rho_new/rho                  (COCG)
    __shared__ cplx red[2 * FH_FIN_BLOCK];
    __shared__ int cnt;
    const int node = blockIdx.x, t = threadIdx.x;
    if (a.s.node_active[node] == 0) return;
    if (t == 0) cnt = 0;
    cplx rho_new, rr;
    fh_sum_partials2<LD>(a.partial1, a.partial2, node, a.nblk, red, rho_new, rr);
    if (t < LD) {
        const int i = node * LD + t;
        if (a.s.active[i]) {
            double rn = sqrt(rr.x);
            a.s.rnorm[i] = rn;
            a.s.iters[i] += 1;
            int act = 1;
            // non-finite first: `!(NaN > target)` is true and would mark a NaN residual as converged
            if (!isfinite(rn)) { act = 0; a.s.status[i] = 8; }
            else if (!(rn > a.s.target[i])) { act = 0; a.s.status[i] = 0; }
            else {
                cplx beta = cdiv(rho_new, a.s.rho[i]);
                bool bad = cabs2(a.s.rho[i]) == 0.0;
                if (a.mode == 0) {
                    cplx om = a.s.omega[i];
                    beta = cmul(beta, cdiv(a.s.alpha[i], om));
                    bad = bad || cabs2(om) == 0.0;
                }
                if (bad || !fh_finite(beta)) {
                    act = 0; a.s.status[i] = 8;
                } else {
                    a.s.beta[i] = beta;
                    a.s.rho[i] = rho_new;
                }
            }
            a.s.active[i] = act;
            if (act) atomicAdd(&cnt, 1);
        }
    }
    __syncthreads();
    if (t == 0) a.s.node_active[node] = cnt;
}

// total number of active columns over all nodes -> *out (one block)
__global__ void k_count_active(const int* node_active, int nodes, int* out) {
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < nodes; ++i) s += node_active[i];
        *out = s;
    }
}

// progress word for the host: the Krylov driver never blocks in hipStreamSynchronize while
// iterating (each such round trip idles the GPU for milliseconds); it polls this word instead.
__global__ void k_publish_progress(const int* node_active, int nodes, unsigned long long* progress, unsigned tag) {
    if (threadIdx.x == 0) {
        unsigned s = 0;
        for (int i = 0; i < nodes; ++i) s += (unsigned)node_active[i];
        // relaxed: the word itself is the whole message, no other data is handed to the host
        __hip_atomic_store(progress, ((unsigned long long)tag << 32) | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------
// COCG (conjugate orthogonal CG) for COMPLEX SYMMETRIC shifted systems: real-symmetric (or
// complex-symmetric) A, B with a complex shift give S = zB - A = S^T, so the BiCG recurrences
// collapse to one operator application per iteration with the unconjugated bilinear form.
// Per iteration: 1 SpMM (2 panel passes) + 6 + 3 vector passes, against 2 SpMM + 14 for BiCGStab.
// ------------------------------------------------------------------------------------
template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_cocg_init(fh_vec_args a) {
    // P = R ; partial1 = sum R*R (unconjugated), partial2 = sum |R|^2
    const int node = blockIdx.y;
    const size_t o = ((size_t)node * gridDim.x + blockIdx.x) * LD;
    const size_t total = (size_t)a.N * LD;
    const CT* R = (const CT*)a.R + (size_t)node * a.node_stride;
    CT* P = (CT*)a.P + (size_t)node * a.node_stride;
    cplx d1 = cmake(0, 0), d2 = cmake(0, 0);
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        CT r = R[e];
        P[e] = r;
        const cplx rd = to_d(r);
        d1 = cadd(d1, cmul(rd, rd));
        d2.x += cabs2(rd);
    }
    __shared__ cplx red[FH_BLOCK];
    fh_block_reduce_cols<LD>(d1, red, a.partial1 + o);
    fh_block_reduce_cols<LD>(d2, red, a.partial2 + o);
}

// Sum-mode start from ONE shared source panel: R_node = f_node,c * SRC, P = R, partial1 = sum R*R, partial2 = sum |R|^2
// with f = 1/(z_node - lambda_c) (Ritz warm start: the residual of Y0 = q/(z - lambda) is (A q - lambda B q)/(z - lambda),
// the same vector for every node up to that scalar) or f = 1 (zero guess: R = RHS).  Replaces, per node, the warm-start
// panel, the residual product over it and the separate P = R pass.
template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_cocg_init_shared(fh_vec_args a) {
    const int node = blockIdx.y;
    const size_t o = ((size_t)node * gridDim.x + blockIdx.x) * LD;
    const size_t total = (size_t)a.N * LD;
    const int c = threadIdx.x % LD;
    cplx f = cmake(1, 0);
    if (a.lambda) {
        const cplx z = a.znode[node];
        f = cdiv(cmake(1, 0), cmake(z.x - a.lambda[c], z.y));
    }
    CT* R = (CT*)a.R + (size_t)node * a.node_stride;
    CT* P = (CT*)a.P + (size_t)node * a.node_stride;
    cplx d1 = cmake(0, 0), d2 = cmake(0, 0);
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const cplx rd = cmul(a.Q[e], f);
        const CT r = cvt<CT>(rd);
        R[e] = r;
        P[e] = r;
        d1 = cadd(d1, cmul(rd, rd));
        d2.x += cabs2(rd);
    }
    __shared__ cplx red[FH_BLOCK];
    fh_block_reduce_cols<LD>(d1, red, a.partial1 + o);
    fh_block_reduce_cols<LD>(d2, red, a.partial2 + o);
}

// The same start WITHOUT the panels: r^T r and |r|^2 of R_node = f_node,c * SRC are f^2 sum SRC^2 and |f|^2 sum |SRC|^2 -- one
// pass over the ONE source panel gives the partial rows of every node; R and P are first written by the first fused vector
// kernel (fh_vec_args::first_src), the first operator product reads SRC itself (fh_spmm_args::colscale).  Saves, per sweep,
// 2 x nodes panel writes here, (nodes - 1) panel reads in the first product and 2 x nodes panel reads in the first update.
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_cocg_init_lazy(fh_vec_args a) {
    const size_t total = (size_t)a.N * LD;
    const int c = threadIdx.x % LD;
    cplx d1 = cmake(0, 0), d2 = cmake(0, 0);
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        const cplx v = a.Q[e];
        d1 = cadd(d1, cmul(v, v));
        d2.x += cabs2(v);
    }
    __shared__ cplx red[FH_BLOCK];
    __shared__ cplx tot[2][LD];
    fh_block_reduce_cols<LD>(d1, red, &tot[0][0]);
    fh_block_reduce_cols<LD>(d2, red, &tot[1][0]);
    __syncthreads();
    for (int idx = threadIdx.x; idx < a.nodes * LD; idx += FH_BLOCK) {
        const int n = idx / LD, cc = idx % LD;
        const cplx f = a.first_scale[n * LD + cc];
        const size_t o = ((size_t)n * gridDim.x + blockIdx.x) * LD + cc;
        a.partial1[o] = cmul(cmul(f, f), tot[0][cc]);
        a.partial2[o] = cmake(cabs2(f) * tot[1][cc].x, 0.0);
    }
    (void)c;
}

// Q_proj of a sum-mode sweep: OUT = [Re] ( SRC * rho_c + ACC ), rho_c = sum_e w_e / (z_e - lambda_c) -- the weighted sum of
// the warm starts in closed form (no per-node panels) plus the accumulated Krylov corrections.  rho == null: OUT = [Re] ACC.
template <int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_sum_finish(const cplx* __restrict__ src, const cplx* __restrict__ rho,
                                                          const cplx* __restrict__ acc, cplx* __restrict__ out, size_t total, int real_part) {
    const cplx rc = rho ? rho[threadIdx.x % LD] : cmake(0, 0);
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
        cplx v = acc[e];
        if (rho) cfma(v, rc, src[e]);
        if (real_part) v.y = 0.0;
        out[e] = v;
    }
}

template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_cocg_update(fh_vec_args a) {
    // X += alpha P ; R -= alpha Q (Q stored in V) ; partial1 = sum R*R, partial2 = sum |R|^2
    // sum mode: X is not touched here (k_cocg_p_sum adds alpha P to the shared accumulator): 3 passes
    const int node = blockIdx.y;
    const int c = threadIdx.x % LD;
    if (a.counters && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.counters + 2, (unsigned long long)a.s.node_active[node]);
    const size_t o = ((size_t)node * gridDim.x + blockIdx.x) * LD;
    cplx d1 = cmake(0, 0), d2 = cmake(0, 0);
    const bool on = a.s.node_active[node] != 0 && a.s.active[node * LD + c];
    if (on) {
        const size_t total = (size_t)a.N * LD;
        const CT alpha = cvt<CT>(a.s.alpha[node * LD + c]);
        const CT* Q = (const CT*)a.V + (size_t)node * a.node_stride;
        CT* R = (CT*)a.R + (size_t)node * a.node_stride;
        if (a.sum_acc) {
            for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
                CT r = csub(R[e], cmul(alpha, Q[e]));
                R[e] = r;
                const cplx rd = to_d(r);
                d1 = cadd(d1, cmul(rd, rd));
                d2.x += cabs2(rd);
            }
        } else {
            const CT* P = (const CT*)a.P + (size_t)node * a.node_stride;
            CT* X = (CT*)a.X + (size_t)node * a.node_stride;
            for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK) {
                CT x = X[e];
                cfma(x, alpha, P[e]);
                X[e] = x;
                CT r = csub(R[e], cmul(alpha, Q[e]));
                R[e] = r;
                const cplx rd = to_d(r);
                d1 = cadd(d1, cmul(rd, rd));
                d2.x += cabs2(rd);
            }
        }
    }
    __shared__ cplx red[FH_BLOCK];
    fh_block_reduce_cols<LD>(d1, red, a.partial1 + o);
    fh_block_reduce_cols<LD>(d2, red, a.partial2 + o);
}

template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_cocg_p(fh_vec_args a) {
    // P = R + beta P
    const int node = blockIdx.y;
    if (a.s.node_active[node] == 0) return;
    const int c = threadIdx.x % LD;
    if (!a.s.active[node * LD + c]) return;
    const size_t total = (size_t)a.N * LD;
    const CT beta = cvt<CT>(a.s.beta[node * LD + c]);
    const CT* R = (const CT*)a.R + (size_t)node * a.node_stride;
    CT* P = (CT*)a.P + (size_t)node * a.node_stride;
    for (size_t e = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * FH_BLOCK)
        P[e] = cadd(R[e], cmul(beta, P[e]));
}

// Sum mode: P = R + beta P for every node AND  ACC += sum_nodes w_node alpha [scale] P_old  in one
// pass.  contour_apply only needs Q_proj = sum_e w_e Y_e, never Y_e itself, so the solution panels
// (one read + one write per node and iteration in k_cocg_update) are replaced by one shared
// accumulator (one read + one write per iteration for ALL nodes).  Nodes are summed in index
// order by the thread that owns the element: deterministic, no atomics.  A column that converged
// in this iteration is no longer `active` but still has `accum` set (fin_alpha): its last step is
// accumulated, its P is left alone.
#define FH_PSUM_K 4
template <typename CT, int LD>
__global__ __launch_bounds__(FH_BLOCK) void k_cocg_p_sum(fh_vec_args a) {
    const size_t total = (size_t)a.N * LD;
    const size_t e0 = (size_t)blockIdx.x * FH_PSUM_K * FH_BLOCK + threadIdx.x;
    const int c = threadIdx.x % LD;
    cplx acc[FH_PSUM_K];
#pragma unroll
    for (int k = 0; k < FH_PSUM_K; ++k) acc[k] = cmake(0, 0);
    bool any = false;
    for (int n = 0; n < a.nodes; ++n) {
        if (!a.s.node_accum[n]) continue;                       // uniform over the grid
        const int i = n * LD + c;
        if (!a.s.accum[i]) continue;
        any = true;
        const bool act = a.s.active[i] != 0;
        cplx coef = cmul(a.wnode[n], a.s.alpha[i]);
        if (a.sum_scale) { const double sc = a.sum_scale[i]; coef.x *= sc; coef.y *= sc; }
        const CT beta = cvt<CT>(a.s.beta[i]);
        const CT* R = (const CT*)a.R + (size_t)n * a.node_stride;
        CT* P = (CT*)a.P + (size_t)n * a.node_stride;
#pragma unroll
        for (int k = 0; k < FH_PSUM_K; ++k) {
            const size_t e = e0 + (size_t)k * FH_BLOCK;
            if (e < total) {
                const CT p = P[e];
                cfma(acc[k], coef, to_d(p));
                if (act) P[e] = cadd(R[e], cmul(beta, p));
            }
        }
    }
    if (any) {
#pragma unroll
        for (int k = 0; k < FH_PSUM_K; ++k) {
            const size_t e = e0 + (size_t)k * FH_BLOCK;
            if (e < total) a.sum_acc[e] = cadd(a.sum_acc[e], acc[k]);
        }
    }
}

// ------------------------------------------------------------------------------------
// Fused COCG iteration: three launches per iteration (SpMM, one finalize, one vector kernel) instead of five, one global
// reduction point instead of two, and one panel pass fewer (the residual is read and written once per iteration, not twice).
//
//   k_spmm (dot_mode 6)   q = S p,  sigma = p^T q,  kappa = q^T q
//   k_fused_fin           alpha = rho / sigma with the TRUE rho = r^T r (summed from the previous vector kernel's partials);
//                         rho of the NEXT residual is predicted from the recurrence r' = r - alpha q and the conjugacy
//                         r^T q = p^T q:      rho' = alpha^2 kappa - rho,      beta = rho' / rho
//   k_fused_vec           r -= alpha q;  ACC += w alpha p (sum mode) or x += alpha p;  p = r + beta p;  partials of the
//                         TRUE r^T r and |r|^2 of the new residual for the next finalize
//
// Why this is safe: x += alpha p, r -= alpha (S p) keeps r = b - S x for ANY alpha and beta, so the residual that is tested
// is always the residual of the iterate; alpha uses the true rho, only beta sees the prediction, and an inexact beta costs
// conjugacy, not correctness.  (Measured before building it, numpy on 14 400-unknown pencils: iterations to 3e-2 / 1e-4 /
// 1e-8 / 1e-12 are 133 / 299 / 381 / 480 with the true beta and 133 / 299 / 391 / 489 with the predicted one on the
// hardest circle node, identical on every other node and on the tall ellipse.)  A first version streamed r through the
// SpMM as well (r^T q, q^H r, |q|^2: exact rho' and the norm of r' one step ahead): the extra pass cost the gather kernel
// 34 us per launch, more than the two launches it removed.
// The stop test runs on the TRUE norm of the previous vector kernel, i.e. one SpMM late.  With inexact solves
// (rtol >= 1e-3) a column also stops when the predicted |rho'| -- scaled by the current ratio |r|^2 / |r^T r| -- is below
// target^2.  That estimate of |r'|^2 is exact when the residuals are complex multiples of real vectors (real right-hand
// side and B = I or B a polynomial in A: cfg 3), and was measured within a factor 0.5 .. 2.4 (median 1.00) of the true
// norm on a pencil whose A and B do not commute: it saves the SpMM of the last iteration of a node (10 % of the SpMM
// node passes of a cfg-3 solve) at the price of stopping a few columns at up to twice the inexact target.
// ------------------------------------------------------------------------------------
#define FH_FV_THREADS 512
#define FH_FV_EMAX 28

// half != 0: the geometry of the launches that own FH_FV_EMAX / 2 elements per thread (complex64 panels; the lazy start's first launch)
void fh_fused_vec_geometry(int N, int ld, int half, int* nblk, int* nseg, int* per_thread) {
    const size_t total = (size_t)N * ld;
    const size_t emax = half ? FH_FV_EMAX / 2 : FH_FV_EMAX;          // elements a thread owns (see k_fused_vec)
    size_t g = (total + FH_FV_THREADS - 1) / FH_FV_THREADS;
    if (g > 256) g = 256;                              // one 512-thread workgroup per CU (2 waves/SIMD, 256 VGPRs): 256 partial rows per node and segment
    if (g < 1) g = 1;
    size_t e = (total + g * FH_FV_THREADS - 1) / (g * FH_FV_THREADS);
    size_t segs = 1;
    if (e > emax) { segs = (e + emax - 1) / emax; e = (total + g * FH_FV_THREADS * segs - 1) / (g * FH_FV_THREADS * segs); }
    *nblk = (int)g; *nseg = (int)segs; *per_thread = (int)e;
}

template <int LD>
__global__ __launch_bounds__(FH_FIN_BLOCK) void k_fused_fin(fh_fused_fin_args a) {
    // grid (LD / 16, nodes): one workgroup sums the partial rows of one 16-column tile of one node.  64 row groups x 16
    // columns; every thread has all its loads in flight at once (the rows were written by other XCDs: each load is a
    // trip to the Infinity Cache), summed in a fixed order.
    __shared__ double red[FH_FIN_BLOCK / 64][16][8];
    __shared__ int cnt_active, cnt_accum;
    const int node = blockIdx.y, tile = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int c16 = lane & 15, g = wave * 4 + (lane >> 4);   // 64 row groups x 16 columns
    const int c = tile * 16 + c16;
    const int i = node * LD + c;
    if (a.s.node_active[node] == 0) {                   // (node_active is only rewritten by the node's LAST tile workgroup, at its end)
        if (!a.final_check) {
            if (t < 16) a.s.accum[i] = 0;
            if (tile == 0 && t == 0) a.s.node_accum[node] = 0;
        }
        return;
    }
    if (t == 0) { cnt_active = 0; cnt_accum = 0; }
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = 0.0;
    {
        const size_t base = (size_t)node * a.nblk_vec * LD + c;
#pragma unroll 4
        for (int b = g; b < a.nblk_vec; b += 64) {
            const cplx r1 = a.rho[base + (size_t)b * LD], r2 = a.rr[base + (size_t)b * LD];
            v[0] += r1.x; v[1] += r1.y; v[2] += r2.x;
        }
    }
    if (!a.final_check) {
        const size_t base = (size_t)node * a.nblk_op * LD + c;
#pragma unroll 4
        for (int b = g; b < a.nblk_op; b += 64) {
            const size_t o = base + (size_t)b * LD;
            const cplx s1 = a.sig[o], s3 = a.kap[o];
            v[3] += s1.x; v[4] += s1.y; v[5] += s3.x; v[6] += s3.y;
        }
    }
    // the four row groups of a wave (lanes c16, c16 + 16, + 32, + 48), then the 16 waves through LDS in a fixed order
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        v[q] += __shfl_xor(v[q], 16);
        v[q] += __shfl_xor(v[q], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int q = 0; q < 7; ++q) red[wave][lane][q] = v[q];
    }
    __syncthreads();
    if (t < 16) {
        double w[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            double sacc = red[0][t][q];
#pragma unroll
            for (int k = 1; k < FH_FIN_BLOCK / 64; ++k) sacc += red[k][t][q];
            w[q] = sacc;
        }
        int stepped = 0;
        if (a.s.active[i]) {
            const cplx rho = cmake(w[0], w[1]);
            const double rr = w[2];
            const double rn = sqrt(rr);
            a.s.rnorm[i] = rn;
            a.s.rho[i] = rho;
            int act = 1;
            // non-finite first: `!(NaN > target)` is true and would mark a NaN residual as converged
            if (!isfinite(rn)) { act = 0; a.s.status[i] = 8; }
            else if (!(rn > a.s.target[i])) { act = 0; a.s.status[i] = 0; }       // the true norm says: converged
            else if (!a.final_check) {
                const cplx sigma = cmake(w[3], w[4]), kappa = cmake(w[5], w[6]);
                const cplx al = cdiv(rho, sigma);
                if (cabs2(sigma) == 0.0 || cabs2(rho) == 0.0 || !fh_finite(al)) {
                    act = 0; a.s.status[i] = 8;                                   // breakdown
                } else {
                    const cplx rho_next = csub(cmul(cmul(al, al), kappa), rho);
                    const cplx beta = cdiv(rho_next, rho);
                    a.s.alpha[i] = al;
                    a.s.iters[i] += 1;
                    stepped = 1;
                    const double tg = a.s.target[i];
                    // |r'|^2 estimated as |rho'| * (|r|^2 / |r^T r|); inexact mode only (see the header of this section)
                    const double rr_next = sqrt(cabs2(rho_next)) * (rr / sqrt(cabs2(rho)));
                    if (a.predict_stop && rr_next <= tg * tg && rr_next >= 1e-12 * rr) {
                        act = 0; a.s.status[i] = 0;
                        a.s.rnorm[i] = sqrt(rr_next);
                        a.s.beta[i] = cmake(0, 0);
                    } else if (!fh_finite(beta)) {
                        act = 0; a.s.status[i] = 8;
                        a.s.beta[i] = cmake(0, 0);
                    } else {
                        a.s.beta[i] = beta;
                    }
                }
            }
            a.s.active[i] = act;
            if (act) atomicAdd(&cnt_active, 1);
        }
        if (!a.final_check) {
            a.s.accum[i] = stepped;
            if (stepped) atomicAdd(&cnt_accum, 1);
        }
    }
    __syncthreads();
    // Per-node counters are sums over the LD / 16 tile workgroups: ONE returning 64-bit atomic per tile carries
    // (active columns << 40 | stepping columns << 20 | 1 ticket); the workgroup that draws the last ticket knows both
    // totals from the value returned and publishes node_active / node_accum (read by the NEXT kernels only).  Integer
    // arithmetic: the order of arrival does not matter.
    if (t == 0) {
        unsigned long long* scratch = a.tickets + node;
        const unsigned long long mine = ((unsigned long long)cnt_active << 40) | ((unsigned long long)cnt_accum << 20) | 1ull;
        const unsigned long long old = atomicAdd(scratch, mine);
        if ((old & 0xFFFFFull) == (unsigned long long)gridDim.x - 1ull) {
            const unsigned long long tot = old + mine;
            a.s.node_active[node] = (int)(tot >> 40);
            if (!a.final_check) a.s.node_accum[node] = (int)((tot >> 20) & 0xFFFFFull);
            atomicExch(scratch, 0ull);
        }
    }
}

// FIRST: the lazy start's first iteration -- residual and direction of node n are first_scale[n][c] * first_src (shared by
// the nodes: the reads of the 51 MB source panel are served by the caches after the first node), R and P are written here
template <typename CT, int LD, bool SUM, bool FIRST>
__global__ __launch_bounds__(FH_FV_THREADS, 2) void k_fused_vec(fh_vec_args a, int per_thread) {
    // grid (nblk, nseg).  A thread OWNS per_thread (<= 28) elements of the panel, G * 512 apart (a multiple of LD: one
    // column per thread), for every node: the weighted steps of all nodes are summed in registers and the shared
    // accumulator is read and written once per launch.  One workgroup per CU at two waves per SIMD: the register file
    // holds the 28 accumulators beside twelve 16-byte loads in flight per thread (96 KiB per CU, three times what the
    // HBM latency needs).
    __shared__ double red[FH_FV_THREADS / 64][LD][3];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = t % LD;
    const size_t total = (size_t)a.N * LD;
    const size_t stride = (size_t)gridDim.x * FH_FV_THREADS;
    const size_t e0 = (size_t)blockIdx.y * stride * per_thread + (size_t)blockIdx.x * FH_FV_THREADS + t;
    const int prow = blockIdx.y * gridDim.x + blockIdx.x, nprow = gridDim.x * gridDim.y;
    constexpr int GRP = sizeof(CT) == 8 ? 2 : 4;               // elements per thread in flight at once, 3 loads each (complex64 panels with 4: the compiler spills 90 VGPRs)
    // complex64 panels: half as many elements per thread (and twice the segments) -- with 28 the compiler hoists 3 x 28 64-bit
    // address pairs out of the node loop beside the 112 accumulator registers and spills 74 VGPRs (300 B of scratch per lane)
    // (the lazy start's first launch likewise: its source-panel loads are one more operand per element -- 152 VGPRs spilled
    //  at 28 elements per thread)
    constexpr int EMAX = (sizeof(CT) == 8 || FIRST) ? FH_FV_EMAX / 2 : FH_FV_EMAX;
    cplx acc[EMAX];
#pragma unroll
    for (int j = 0; j < EMAX; ++j) acc[j] = cmake(0, 0);
    bool any = false;
    for (int n = 0; n < a.nodes; ++n) {
        if (!a.s.node_accum[n]) continue;                       // uniform over the grid: nobody writes a partial row
        const int i = n * LD + c;
        const bool step = a.s.accum[i] != 0;
        const bool act = a.s.active[i] != 0;
        if (a.counters && blockIdx.x == 0 && blockIdx.y == 0 && t == 0) {
            atomicAdd(a.counters + 2, (unsigned long long)a.s.node_accum[n]);          // columns stepping at this node
            atomicAdd(a.counters + 3, (unsigned long long)a.s.node_active[n]);         // ... that go on iterating (5 panel passes each)
            if (FIRST) atomicAdd(a.counters + 5, (unsigned long long)a.s.node_active[n]);   // ... in a lazy start's first launch (4: q and the source read, r and p written)
        }
        double d1x = 0.0, d1y = 0.0, d2 = 0.0;
        if (step) {
            any = true;
            const CT alpha = cvt<CT>(a.s.alpha[i]);
            const CT beta = cvt<CT>(a.s.beta[i]);
            cplx coef = cmake(0, 0);
            if (SUM) {
                coef = cmul(a.wnode[n], a.s.alpha[i]);
                if (a.sum_scale) { const double sc = a.sum_scale[i]; coef.x *= sc; coef.y *= sc; }
            }
            const cplx fsc = FIRST ? a.first_scale[i] : cmake(1, 0);
            const cplx* __restrict__ fsrc = a.first_src;
            const CT* __restrict__ Q = (const CT*)a.V + (size_t)n * a.node_stride;
            CT* __restrict__ R = (CT*)a.R + (size_t)n * a.node_stride;
            CT* __restrict__ P = (CT*)a.P + (size_t)n * a.node_stride;
            CT* __restrict__ X = SUM ? nullptr : (CT*)a.X + (size_t)n * a.node_stride;
#pragma unroll
            for (int j0 = 0; j0 < EMAX; j0 += GRP) {
                if (j0 >= per_thread) continue;                      // (no break: the loop must unroll fully -- acc[] lives in registers)
                CT pv[GRP], qv[GRP], rv[GRP], xv[GRP];
                bool ok[GRP];
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const size_t e = e0 + (size_t)(j0 + u) * stride;
                    ok[u] = (j0 + u < per_thread) && e < total;
                    pv[u] = fh_czero<CT>(); qv[u] = fh_czero<CT>(); rv[u] = fh_czero<CT>(); xv[u] = fh_czero<CT>();
                    if (ok[u]) {
                        if (FIRST) { pv[u] = cvt<CT>(cmul(fsc, fsrc[e])); rv[u] = pv[u]; }
                        else pv[u] = P[e];
                        if (act) { qv[u] = fh_ld_nt(Q + e); if (!FIRST) rv[u] = R[e]; }
                        if (!SUM) xv[u] = X[e];
                    }
                }
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const size_t e = e0 + (size_t)(j0 + u) * stride;
                    if (SUM) cfma(acc[j0 + u], coef, to_d(pv[u]));
                    else if (ok[u]) { cfma(xv[u], alpha, pv[u]); X[e] = xv[u]; }
                    if (act && ok[u]) {
                        const CT rn = csub(rv[u], cmul(alpha, qv[u]));
                        R[e] = rn;
                        P[e] = cadd(rn, cmul(beta, pv[u]));
                        const cplx rd = to_d(rn);
                        d1x += rd.x * rd.x - rd.y * rd.y; d1y += 2.0 * rd.x * rd.y; d2 += cabs2(rd);
                    }
                }
            }
        }
        // per-column sums of the workgroup: lanes that share a column inside a wave first (LD < 64), then the 8 waves in
        // a fixed order through LDS
        if (LD < 64) {
#pragma unroll
            for (int off = LD; off < 64; off <<= 1) {
                d1x += __shfl_xor(d1x, off); d1y += __shfl_xor(d1y, off); d2 += __shfl_xor(d2, off);
            }
        }
        if (lane < LD) { red[wave][lane][0] = d1x; red[wave][lane][1] = d1y; red[wave][lane][2] = d2; }
        __syncthreads();
        if (t < LD) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int w = 0; w < FH_FV_THREADS / 64; ++w) { s0 += red[w][t][0]; s1 += red[w][t][1]; s2 += red[w][t][2]; }
            const size_t o = ((size_t)n * nprow + prow) * LD + t;
            a.partial1[o] = cmake(s0, s1);
            a.partial2[o] = cmake(s2, 0.0);
        }
        __syncthreads();
    }
    if (SUM && any && a.counters && blockIdx.x == 0 && blockIdx.y == 0 && t < LD) atomicAdd(a.counters + 4, 1ull);   // accumulator columns touched
    if (SUM && any) {
#pragma unroll
        for (int j = 0; j < EMAX; ++j) {
            const size_t e = e0 + (size_t)j * stride;
            if (j < per_thread && e < total) a.sum_acc[e] = cadd(a.sum_acc[e], acc[j]);
        }
    }
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
#define FH_DISPATCH_VEC(prec, ld, KERNEL, grid, st, args)                                          \
    do {                                                                                            \
        if ((prec) == 32) {                                                                         \
            if ((ld) == 16) hipLaunchKernelGGL((KERNEL<cplxf, 16>), grid, dim3(FH_BLOCK), 0, st, args); \
            else if ((ld) == 32) hipLaunchKernelGGL((KERNEL<cplxf, 32>), grid, dim3(FH_BLOCK), 0, st, args); \
            else hipLaunchKernelGGL((KERNEL<cplxf, 64>), grid, dim3(FH_BLOCK), 0, st, args);        \
        } else {                                                                                    \
            if ((ld) == 16) hipLaunchKernelGGL((KERNEL<cplx, 16>), grid, dim3(FH_BLOCK), 0, st, args); \
            else if ((ld) == 32) hipLaunchKernelGGL((KERNEL<cplx, 32>), grid, dim3(FH_BLOCK), 0, st, args); \
            else hipLaunchKernelGGL((KERNEL<cplx, 64>), grid, dim3(FH_BLOCK), 0, st, args);         \
        }                                                                                           \
    } while (0)

#define FH_DISPATCH_FIN(ld, KERNEL, grid, st, args)                                           \
    do {                                                                                        \
        if ((ld) == 16) hipLaunchKernelGGL((KERNEL<16>), grid, dim3(FH_FIN_BLOCK), 0, st, args); \
        else if ((ld) == 32) hipLaunchKernelGGL((KERNEL<32>), grid, dim3(FH_FIN_BLOCK), 0, st, args); \
        else hipLaunchKernelGGL((KERNEL<64>), grid, dim3(FH_FIN_BLOCK), 0, st, args);           \
    } while (0)

void fh_launch_init_guess(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(64, ld, k_init_guess, dim3(nblk, nodes), st, a);
}
void fh_launch_copy_r(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(a.prec, ld, k_copy_r, dim3(nblk, nodes), st, a);
}
void fh_launch_p_update(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(a.prec, ld, k_p_update, dim3(nblk, nodes), st, a);
}
void fh_launch_s_update(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(a.prec, ld, k_s_update, dim3(nblk, nodes), st, a);
}
void fh_launch_xr_update(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(a.prec, ld, k_xr_update, dim3(nblk, nodes), st, a);
}
void fh_launch_cocg_init(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(a.prec, ld, k_cocg_init, dim3(nblk, nodes), st, a);
}
void fh_launch_cocg_init_shared(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(a.prec, ld, k_cocg_init_shared, dim3(nblk, nodes), st, a);
}
void fh_launch_cocg_init_lazy(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    (void)nodes;
    if (ld == 16) hipLaunchKernelGGL((k_cocg_init_lazy<16>), dim3(nblk), dim3(FH_BLOCK), 0, st, a);
    else if (ld == 32) hipLaunchKernelGGL((k_cocg_init_lazy<32>), dim3(nblk), dim3(FH_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_cocg_init_lazy<64>), dim3(nblk), dim3(FH_BLOCK), 0, st, a);
}
void fh_launch_sum_finish(const cplx* src, const cplx* rho, const cplx* acc, cplx* out, int N, int ld, int real_part, hipStream_t st) {
    const size_t total = (size_t)N * ld;
    const int nblk = (int)std::min<size_t>((total + FH_BLOCK - 1) / FH_BLOCK, 2048);
    if (ld == 16) hipLaunchKernelGGL((k_sum_finish<16>), dim3(nblk), dim3(FH_BLOCK), 0, st, src, rho, acc, out, total, real_part);
    else if (ld == 32) hipLaunchKernelGGL((k_sum_finish<32>), dim3(nblk), dim3(FH_BLOCK), 0, st, src, rho, acc, out, total, real_part);
    else hipLaunchKernelGGL((k_sum_finish<64>), dim3(nblk), dim3(FH_BLOCK), 0, st, src, rho, acc, out, total, real_part);
}
void fh_launch_cocg_update(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(a.prec, ld, k_cocg_update, dim3(nblk, nodes), st, a);
}
void fh_launch_cocg_p(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st) {
    FH_DISPATCH_VEC(a.prec, ld, k_cocg_p, dim3(nblk, nodes), st, a);
}
void fh_launch_cocg_p_sum(const fh_vec_args& a, int ld, int nodes, hipStream_t st) {
    (void)nodes;
    const size_t total = (size_t)a.N * ld;
    const int nblk = (int)((total + (size_t)FH_PSUM_K * FH_BLOCK - 1) / ((size_t)FH_PSUM_K * FH_BLOCK));
    FH_DISPATCH_VEC(a.prec, ld, k_cocg_p_sum, dim3(nblk), st, a);
}
void fh_launch_fin_init(const fh_fin_args& a, int ld, int nodes, hipStream_t st) {
    FH_DISPATCH_FIN(ld, k_fin_init, dim3(nodes), st, a);
}
void fh_launch_fin_alpha(const fh_fin_args& a, int ld, int nodes, hipStream_t st) {
    FH_DISPATCH_FIN(ld, k_fin_alpha, dim3(nodes), st, a);
}
void fh_launch_fin_omega(const fh_fin_args& a, int ld, int nodes, hipStream_t st) {
    FH_DISPATCH_FIN(ld, k_fin_omega, dim3(nodes), st, a);
}
void fh_launch_fin_rho(const fh_fin_args& a, int ld, int nodes, hipStream_t st) {
    FH_DISPATCH_FIN(ld, k_fin_rho, dim3(nodes), st, a);
}
void fh_launch_count_active(const int* node_active, int nodes, int* out, hipStream_t st) {
    hipLaunchKernelGGL(k_count_active, dim3(1), dim3(64), 0, st, node_active, nodes, out);
}
void fh_launch_publish_progress(const int* node_active, int nodes, unsigned long long* progress, unsigned tag, hipStream_t st) {
    hipLaunchKernelGGL(k_publish_progress, dim3(1), dim3(64), 0, st, node_active, nodes, progress, tag);
}
void fh_launch_narrow_scaled(const cplx* src, size_t src_stride, cplxf* dst, size_t dst_stride, const double* r0norm,
                             int N, int ld, int nblk, int nodes, hipStream_t st) {
    size_t total = (size_t)N * ld;
    if (ld == 16) hipLaunchKernelGGL((k_narrow_scaled<16>), dim3(nblk, nodes), dim3(FH_BLOCK), 0, st, src, src_stride, dst, dst_stride, r0norm, total);
    else if (ld == 32) hipLaunchKernelGGL((k_narrow_scaled<32>), dim3(nblk, nodes), dim3(FH_BLOCK), 0, st, src, src_stride, dst, dst_stride, r0norm, total);
    else hipLaunchKernelGGL((k_narrow_scaled<64>), dim3(nblk, nodes), dim3(FH_BLOCK), 0, st, src, src_stride, dst, dst_stride, r0norm, total);
}
void fh_launch_widen_axpy(cplx* X, size_t x_stride, const cplxf* D, size_t d_stride, const double* r0norm,
                          int N, int ld, int nblk, int nodes, hipStream_t st) {
    size_t total = (size_t)N * ld;
    if (ld == 16) hipLaunchKernelGGL((k_widen_axpy<16>), dim3(nblk, nodes), dim3(FH_BLOCK), 0, st, X, x_stride, D, d_stride, r0norm, total);
    else if (ld == 32) hipLaunchKernelGGL((k_widen_axpy<32>), dim3(nblk, nodes), dim3(FH_BLOCK), 0, st, X, x_stride, D, d_stride, r0norm, total);
    else hipLaunchKernelGGL((k_widen_axpy<64>), dim3(nblk, nodes), dim3(FH_BLOCK), 0, st, X, x_stride, D, d_stride, r0norm, total);
}

void fh_launch_fused_fin(const fh_fused_fin_args& a, int ld, int nodes, hipStream_t st) {
    if (ld == 16) hipLaunchKernelGGL((k_fused_fin<16>), dim3(1, nodes), dim3(FH_FIN_BLOCK), 0, st, a);
    else if (ld == 32) hipLaunchKernelGGL((k_fused_fin<32>), dim3(2, nodes), dim3(FH_FIN_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_fused_fin<64>), dim3(4, nodes), dim3(FH_FIN_BLOCK), 0, st, a);
}
template <typename CT, int LD>
static void launch_fused_vec_t(const fh_vec_args& a, dim3 grid, int per_thread, hipStream_t st) {
    if (a.first_src) {       // lazy start: sum mode, fp64 panels only (fh_krylov)
        if (a.sum_acc) hipLaunchKernelGGL((k_fused_vec<CT, LD, true, true>), grid, dim3(FH_FV_THREADS), 0, st, a, per_thread);
        else hipLaunchKernelGGL((k_fused_vec<CT, LD, false, true>), grid, dim3(FH_FV_THREADS), 0, st, a, per_thread);
        return;
    }
    if (a.sum_acc) hipLaunchKernelGGL((k_fused_vec<CT, LD, true, false>), grid, dim3(FH_FV_THREADS), 0, st, a, per_thread);
    else hipLaunchKernelGGL((k_fused_vec<CT, LD, false, false>), grid, dim3(FH_FV_THREADS), 0, st, a, per_thread);
}
void fh_launch_fused_vec(const fh_vec_args& a, int ld, hipStream_t st) {
    int nblk, nseg, per;
    fh_fused_vec_geometry(a.N, ld, a.prec == 32 || a.first_src != nullptr, &nblk, &nseg, &per);
    const dim3 grid(nblk, nseg);
    if (a.prec == 32) {
        if (ld == 16) launch_fused_vec_t<cplxf, 16>(a, grid, per, st);
        else if (ld == 32) launch_fused_vec_t<cplxf, 32>(a, grid, per, st);
        else launch_fused_vec_t<cplxf, 64>(a, grid, per, st);
    } else {
        if (ld == 16) launch_fused_vec_t<cplx, 16>(a, grid, per, st);
        else if (ld == 32) launch_fused_vec_t<cplx, 32>(a, grid, per, st);
        else launch_fused_vec_t<cplx, 64>(a, grid, per, st);
    }
}
