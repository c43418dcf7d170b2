// fh_common.hpp -- shared types for libfeasthip (gfx950 / CDNA4 only).
//
// Internal block-vector layout ("panel"): an N x m complex block is stored ROW-MAJOR with a
// padded row length ld in {16,32,64} c128 elements, so one row of a 64-column block is one
// contiguous 1 KiB line = one wave-wide 16 B/lane access.  Lane <-> column, so every
// per-column scalar of the batched Krylov solver (alpha_c, omega_c, ...) lives in the lane
// that owns column c, and a gathered SpMM row read is a single coalesced wave access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>

struct __attribute__((aligned(16))) cplx { double x, y; };

__host__ __device__ inline cplx cmake(double a, double b) { cplx r; r.x = a; r.y = b; return r; }
__host__ __device__ inline cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cplx cmul(cplx a, cplx b) { return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__host__ __device__ inline cplx cmulc(cplx a, cplx b) { /* conj(a)*b */ return cmake(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }
__host__ __device__ inline cplx cscale(cplx a, double s) { return cmake(a.x * s, a.y * s); }
__host__ __device__ inline cplx cconj(cplx a) { return cmake(a.x, -a.y); }
__host__ __device__ inline double cabs2(cplx a) { return a.x * a.x + a.y * a.y; }
__host__ __device__ inline cplx cdiv(cplx a, cplx b) {
    double d = b.x * b.x + b.y * b.y;
    return cmake((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}
// acc += a*b
__host__ __device__ inline void cfma(cplx& acc, cplx a, cplx b) {
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
}
// matrix value (real or complex) times complex
__host__ __device__ inline cplx vmul(double a, cplx b) { return cmake(a * b.x, a * b.y); }
__host__ __device__ inline cplx vmul(cplx a, cplx b) { return cmul(a, b); }

// ---- single-precision complex for the mixed-precision Krylov correction solves -------------
struct __attribute__((aligned(8))) cplxf { float x, y; };
__host__ __device__ inline cplxf cmakef(float a, float b) { cplxf r; r.x = a; r.y = b; return r; }
__host__ __device__ inline cplxf cadd(cplxf a, cplxf b) { return cmakef(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cplxf csub(cplxf a, cplxf b) { return cmakef(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cplxf cmul(cplxf a, cplxf b) { return cmakef(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__host__ __device__ inline void cfma(cplxf& acc, cplxf a, cplxf b) {
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
}
__host__ __device__ inline cplxf vmul(double a, cplxf b) { float f = (float)a; return cmakef(f * b.x, f * b.y); }
__host__ __device__ inline cplxf vmul(cplx a, cplxf b) { return cmul(cmakef((float)a.x, (float)a.y), b); }
// conversions: to_d widens to double complex (all reductions run in fp64); cvt<CT> narrows
__host__ __device__ inline cplx to_d(cplx a) { return a; }
__host__ __device__ inline cplx to_d(cplxf a) { return cmake((double)a.x, (double)a.y); }
template <typename CT> __host__ __device__ inline CT cvt(cplx a);
template <> __host__ __device__ inline cplx cvt<cplx>(cplx a) { return a; }
template <> __host__ __device__ inline cplxf cvt<cplxf>(cplx a) { return cmakef((float)a.x, (float)a.y); }

#define FH_MAX_LD 64

#define FH_CHECK(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            h->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);              \
            return (_e == hipErrorOutOfMemory) ? 6 : 7;                                     \
        }                                                                                   \
    } while (0)

// ---- profiling of kernel classes with HIP events on the handle's stream -----------------
struct fh_prof_class {
    double total_ms = 0.0;
    int64_t launches = 0;
};

struct fh_event_pair {
    hipEvent_t a, b;
    std::string cls;
};

// ---- device CSR (union pattern of A and B) ---------------------------------------------
struct fh_csr {
    int64_t N = 0, nnz = 0;
    int is_complex = 0;      // values are double (0) or cplx (1)
    int b_identity = 0;      // no B values: B = I
    int* rowptr = nullptr;   // N+1
    int* col = nullptr;      // nnz
    void* aval = nullptr;    // nnz x (double|cplx), A on the union pattern
    void* bval = nullptr;    // nnz x (double|cplx), B on the union pattern (null if identity)
    // rows padded to whole chunks of 8 nonzeros for the row-per-wave SpMM (real values only): chunk range of row i is
    // [rp8[i], rp8[i+1]); col8 / a8 / b8 hold 8 entries per chunk, padding = (own row, 0.0, 0.0)
    int* rp8 = nullptr; int* col8 = nullptr; double* a8 = nullptr; double* b8 = nullptr;
    int* perm = nullptr;     // N: row/column renumbering applied at ingest, perm[internal] = caller's index (null: none)
    // row blocks of the renumbered matrix (LDS-window SpMM): block b = rows [blk_start[b], blk_start[b+1]) (<= FH_SPMM_R),
    // ext_idx[ext_ptr[b] .. ext_ptr[b+1]) = the distinct rows OUTSIDE the block its nonzeros touch (<= FH_SPMM_EXT kept),
    // lcol[k] = LDS slot of nonzero k: row - blk_start[b] inside the block, FH_SPMM_R + position in the block's ext list
    // outside, 0xFFFF when the ext list was full (the kernel then gathers col[k] from global memory)
    int nblk = 0;
    int* blk_start = nullptr;
    int* ext_ptr = nullptr;
    int* ext_idx = nullptr;
    unsigned short* lcol = nullptr;
};
#define FH_SPMM_R 128        // rows per block of the ingest renumbering / the LDS-window SpMM
#define FH_SPMM_EXT 160      // outside rows a block stages next to its own: (128 + 160) x 256 B = 72 KiB of LDS, two blocks per CU

struct fh_dense {
    int64_t N = 0;
    int is_complex = 0;
    int b_identity = 0;
    void* A = nullptr;       // N x N column-major, double or cplx
    void* B = nullptr;
};

// Per-(node,column) Krylov scalars, struct of arrays; each array is [nodes][ld].
struct fh_krylov_scalars {
    cplx* rho = nullptr;
    cplx* alpha = nullptr;
    cplx* omega = nullptr;
    cplx* beta = nullptr;
    double* r0norm = nullptr;
    double* target = nullptr;
    double* rnorm = nullptr;
    int* active = nullptr;    // 1 while the column is still iterating
    int* iters = nullptr;     // iterations performed by the column
    int* status = nullptr;    // 0 converged, 5 not converged, 8 breakdown
    int* node_active = nullptr;  // [nodes]: number of active columns of the node
    int* accum = nullptr;        // sum mode: column took an alpha step in this iteration (set by fin_alpha)
    int* node_accum = nullptr;   // [nodes]: any column of the node did
};

struct fh_comm;   // fh_comm.hip: RCCL (or shared-device) communicator attached by feasthip_comm_init_rank

struct feasthip_ctx {
    int device = 0;
    fh_comm* comm = nullptr;
    int64_t col_block_lo = 0, col_block_hi = -1;   // feasthip_set_column_block: columns this rank sweeps (hi < 0: all)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;

    // problem
    int kind = 0;  // 0 none, 1 dense, 2 sparse
    fh_csr csr;
    fh_dense dense;

    // contour
    std::vector<cplx> zne, wne;
    double weight_scale = 2.0;
    int real_projection = 0;
    int node_first = 0, node_count = 0;
    std::vector<int> node_ids;      // local node -> contour index (set by range or list)

    // solver options
    int solver = 1;
    double rtol = 1e-12, atol = 0.0;
    int maxit = 500, restart = 30, factor_precision = 64, cache_factors = 1;

    std::vector<int> last_col_iters;    // [local node][m] iterations per column of the last sweep
    int last_col_m = 0;
    std::vector<int> last_node_iters;   // per local node: max column iterations of the last sweep
    std::vector<int> global_node_iters; // per CONTOUR node, summed over the ranks by the packed reduce of the last sweep

    // workspace (grown lazily)
    std::map<std::string, std::pair<void*, size_t>> bufs;

    // pinned staging ring of the small host -> device uploads (per-column coefficients, weights): fh_upload_coefs
    char* pin = nullptr; size_t pin_cap = 0, pin_off = 0;

    // resident refinement loop (feasthip_contour_apply_resident / rr_reduce_resident / rr_ritz_resident): the panels of one
    // FEAST loop stay on the device in the kernels' own row-major layout between the calls; pointers into `bufs`, dropped
    // (rs_epoch bumped) whenever the problem changes
    int rs_m = 0, rs_ld = 0;                    // columns / padded row length of the projection panel
    cplx* rs_P = nullptr;                       // Q_proj of the last resident sweep, summed over the ranks
    cplx* rs_basis = nullptr;                   // what the Ritz step multiplies: rs_P, or the orthonormalised panel of the fallback
    std::vector<cplx> rs_T;                     // implicit basis Q_proj D^-1: the m diagonal entries 1 / ||column||; empty = the basis is used as is
    int rs_rank = 0;
    cplx* rs_X = nullptr; int rs_X_m = 0, rs_X_ld = 0;   // Ritz vectors of the last resident Ritz step: the next sweep's subspace
    cplx* rs_R = nullptr;                       // A X - B X diag(lambda) of that step: the next sweep's shared start residual
    std::vector<cplx> rs_R_lambda;              // the lambda it was formed with

    // dense LU cache: per local node factors + pivots
    std::vector<void*> lu_factors;
    std::vector<int*> lu_pivots;
    std::vector<int> lu_valid;
    std::vector<cplx> lu_z;
    int lu_prec = 64;             // element type of the cached dense factors: 64 = complex128, 32 = complex64
    // banded LU (CSR input, FEASTHIP_SOLVER_BANDED): bandwidths of the union pattern, factors per node slot
    int csr_kl = 0, csr_ku = 0;
    // pattern of the matrix as stored on the device (ingest order), kept on the host for the band plan of the direct solver
    std::vector<int> host_rowptr, host_col;
    // band plan (fh_banded.hip): 0 not made, 1 narrow band in stored order (one-workgroup elimination), 2 blocked band LU
    // on the dense kernels in band order (band_perm[band row] = stored row, band_iperm its inverse; both on the device)
    int band_plan = 0, band_kl = 0, band_ku = 0;
    int band_prec = 64;           // element type of the cached band factors (blocked plan): 64 = complex128, 32 = complex64
    int* band_perm = nullptr;
    int* band_iperm = nullptr;
    std::vector<void*> band_factors;
    std::vector<int*> band_pivots;
    std::vector<int> band_valid;
    std::vector<cplx> band_z;
    void* mf = nullptr;           // multifrontal plan of the sparse direct solver (fh_dense.hip: fh_mf_state), band_plan == 3
    std::vector<int> col_mask;    // feasthip_set_column_mask: columns with 0 are not iterated by the Krylov solvers
    int poisoned = 0;             // a Krylov deadline / queue fault returned with kernels possibly still queued: every later call fails fast
    int mask_live = 0;            // set only while a contour_apply call runs: the mask is one-shot and never reaches shifted_solve
    int sum_mode = 1;             // COCG contour_apply accumulates alpha*p into one shared panel (FH_NO_SUM_MODE=1 disables)
    int lu_outer_block = 0;       // FH_LU_KB: outer block column of the two-level LU (multiple of 32); 0 = by size (128, 256 from N = 6144)
    int lu_panel_legacy = 0;      // FH_LU_PANEL_LEGACY=1: per-column global-memory panel kernel
    int lu_solve_legacy = 0;      // FH_LU_SOLVE_32=1: 32-column one-launch substitution steps (comparison)
    int lu_lookahead = 1;         // FH_LU_LOOKAHEAD=0: no overlap of the next block column's panels with the rest of the trailing update
    hipStream_t side_stream = nullptr;            // second stream of the LU look-ahead (created on first use)
    int side_reserve = -1;                        // CUs per XCD the side stream's mask leaves to the main stream
    hipEvent_t lu_ev_next = nullptr, lu_ev_rest = nullptr;
    int lu_gemm_staged = 0;       // FH_LU_GEMM_STAGED=1: trailing update with both panels through LDS (comparison; always for complex64)

    // host-mapped progress word written by the device: (chunk tag << 32) | active columns
    volatile unsigned long long* h_progress = nullptr;   // pinned host view
    unsigned long long* d_progress = nullptr;            // device view of the same word

    // profiling
    unsigned long long* d_counters = nullptr;   // [0] spmm node-launches, [1] spmm column passes
    int profiling = 0;
    std::map<std::string, fh_prof_class> prof;
    std::map<std::string, double> prof_work;    // algorithmic work (flops) issued per class while profiling (dense MFMA classes)
    int prof_period = 0;                        // feasthip_profile_set_period: 0 = default sampling, 1 = every launch
    std::vector<fh_event_pair> pending_events;
    std::vector<hipEvent_t> event_pool;         // recycled profiling events
    int prof_mult = 1;                          // sampling period multiplier, raised when the host cost of sampling shows
    double prof_host_s = 0.0;                   // host seconds spent recording / reading events since profile_enable
    double prof_t0 = 0.0;                       // steady-clock seconds at profile_enable
};

// workspace helper: returns a device buffer of at least `bytes`, reallocating if needed
int fh_get_buf(feasthip_ctx* h, const char* name, size_t bytes, void** out);
void fh_free_bufs(feasthip_ctx* h);

// profiling helpers
void fh_prof_begin(feasthip_ctx* h, const char* cls);
void fh_prof_end(feasthip_ctx* h);
void fh_prof_collect(feasthip_ctx* h);

static inline int fh_pick_ld(int64_t m) {
    if (m <= 16) return 16;
    if (m <= 32) return 32;
    return 64;
}
