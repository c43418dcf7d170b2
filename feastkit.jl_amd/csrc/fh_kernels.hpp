// fh_kernels.hpp -- argument structs and launcher prototypes shared by the .hip units.
#pragma once
#include "fh_common.hpp"

// ---- sparse operator -------------------------------------------------------------------
// Panel pointers are void*: complex128 panels when prec == 64, complex64 when prec == 32.
struct fh_spmm_args {
    const int* rowptr; const int* col; const void* aval; const void* bval;
    int N; int nodes;
    const void* X; size_t x_node_stride;      // element stride between nodes (0 = shared)
    void* Y; size_t y_node_stride;
    const cplx* coefA; const cplx* coefB;     // [nodes][LD] per-column coefficients (always fp64)
    const void* Bvec; size_t b_node_stride;   // non-null: Y = Bvec - S X
    const void* U; size_t u_node_stride;      // dot_mode 1: <U, Y>
    int dot_mode;
    cplx* partial1; cplx* partial2;           // [nodes][nprow][LD], always fp64
    const int* node_active;                   // may be null
    unsigned long long* counters;             // [0] active node-launches, [1] active column x vector passes (may be null)
    int m;                                    // active width when node_active is null
    int uniform_coef;                         // coefA/coefB identical for every column of a node
    int prec;                                 // 64 | 32
    const int* rp8 = nullptr; const int* col8 = nullptr; const double* a8 = nullptr; const double* b8 = nullptr;   // chunk-of-8 rows (fh_csr)
    int use_row_kernel = 0;                   // LD = 64 and real matrix values: k_spmm_row (one wave per row); partial rows = fh_spmm_row_grid(N)
    // k_spmm_row / k_spmm (complex128 panels): X holds a panel SHARED by the nodes whose column c is to be read as colscale[node][c] * X (the lazy
    // start of the sum-mode COCG sweeps: every node's first direction is one source panel times a per-column factor).  The
    // kernel forms y = colscale * (S x) and takes its dots with colscale * x.  Null: X as is.
    const cplx* colscale = nullptr;
    // row blocks of a renumbered matrix (fh_common.hpp: fh_csr); lcol == null: not renumbered, k_spmm serves
    int nblk_rows; const int* blk_start; const int* ext_ptr; const int* ext_idx; const unsigned short* lcol;
};
int fh_spmm_grid(int N, int ld);
int fh_spmm_partials(int N, int ld);
int fh_spmm_row_grid(int N);
int fh_spmm_lds_slots(int nblk_rows, int ld);
void fh_launch_spmm(const fh_spmm_args& a, int ld, bool is_complex, bool bident, int nblk, hipStream_t st);

// ---- Krylov vector kernels (BiCGStab and COCG) -------------------------------------------
struct fh_vec_args {
    int N;
    size_t node_stride;
    void *X, *R, *Rhat, *P, *V, *S, *T;
    const cplx* Q;             // init guess source (shared by all nodes, fp64)
    const double* lambda;      // [LD] Ritz values or null
    const cplx* znode;         // [nodes]
    fh_krylov_scalars s;
    cplx* partial1; cplx* partial2;
    int prec;                  // 64 | 32
    unsigned long long* counters;   // measurement: [2] += active columns of this launch (update kernels), may be null
    // sum mode (COCG inside contour_apply): the per-node solutions are never formed; every step
    // alpha p of every node goes straight into the shared accumulator  ACC += w_node alpha [scale] p
    // lazy start (first fused iteration only): residual and direction of every node are first_scale[node][c] * first_src,
    // the panels R and P do not exist yet -- the kernel reads first_src in their place and WRITES them
    const cplx* first_src;     // N x LD fp64 shared source panel, or null
    const cplx* first_scale;   // [nodes x LD]
    cplx* sum_acc;             // N x LD fp64 accumulator, or null (solutions are updated in X)
    const cplx* wnode;         // [nodes] quadrature weights
    const double* sum_scale;   // [nodes x LD] column scale of the correction (mixed precision) or null
    int nodes;
};
struct fh_fin_args {
    fh_krylov_scalars s;
    const cplx* partial1; const cplx* partial2;
    int nblk; int m;
    double rtol, atol;
    const double* atol_scale;  // per (node,column) factor on atol (mixed precision: 1/||r0||), may be null
    int mode;                  // 0 BiCGStab, 1 COCG
    const int* col_mask;       // [LD] 0 = column keeps its initial guess and is never iterated; may be null
};
void fh_launch_init_guess(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
void fh_launch_copy_r(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
void fh_launch_p_update(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
void fh_launch_s_update(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
void fh_launch_xr_update(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
void fh_launch_cocg_init(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
// sum-mode start from one shared source panel (a.Q = source, a.lambda/a.znode = warm-start factors or null)
void fh_launch_cocg_init_shared(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
// the same start without materialising R and P: only the per-node dot partials (a.Q = source, a.first_scale = per-node column
// factors); the first fused iteration then runs with fh_spmm_args::colscale / fh_vec_args::first_src
void fh_launch_cocg_init_lazy(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
// OUT = [Re](SRC * rho_c + ACC)
void fh_launch_sum_finish(const cplx* src, const cplx* rho, const cplx* acc, cplx* out, int N, int ld, int real_part, hipStream_t st);
void fh_launch_cocg_update(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
void fh_launch_cocg_p(const fh_vec_args& a, int ld, int nblk, int nodes, hipStream_t st);
void fh_launch_cocg_p_sum(const fh_vec_args& a, int ld, int nodes, hipStream_t st);
// ---- fused COCG iteration (fh_sparse.hip, "fused COCG"): SpMM with five dots -> one finalize -> one vector kernel --------
struct fh_fused_fin_args {
    fh_krylov_scalars s;
    const cplx *sig, *kap;                    // SpMM partials [nodes][nblk_op][LD]: p^T q, q^T q (unconjugated)
    const cplx *rho, *rr;                     // partials of the last vector kernel / the init kernel [nodes][nblk_vec][LD]: r^T r, |r|^2
    int nblk_op, nblk_vec;
    unsigned long long* tickets;              // [nodes], zero between launches
    int final_check;                          // 1: only re-evaluate the stop test from (rho, rr) after the last step
    int predict_stop;                         // inexact solves: also stop on the predicted norm of the next residual
};
void fh_launch_fused_fin(const fh_fused_fin_args& a, int ld, int nodes, hipStream_t st);
// geometry of the fused vector kernel for a panel of N x ld elements: blocks per segment, segments, elements per thread
void fh_fused_vec_geometry(int N, int ld, int half, int* nblk, int* nseg, int* per_thread);
// R -= alpha Q, [ACC += w alpha P | X += alpha P], P = R + beta P, partials r^T r and |r|^2 ([nodes][nblk*nseg][LD])
void fh_launch_fused_vec(const fh_vec_args& a, int ld, hipStream_t st);
void fh_launch_fin_init(const fh_fin_args& a, int ld, int nodes, hipStream_t st);
void fh_launch_fin_alpha(const fh_fin_args& a, int ld, int nodes, hipStream_t st);
void fh_launch_fin_omega(const fh_fin_args& a, int ld, int nodes, hipStream_t st);
void fh_launch_fin_rho(const fh_fin_args& a, int ld, int nodes, hipStream_t st);
void fh_launch_count_active(const int* node_active, int nodes, int* out, hipStream_t st);
// writes (tag << 32 | active columns) to a host-mapped word with a system-scope release store
void fh_launch_publish_progress(const int* node_active, int nodes, unsigned long long* progress, unsigned tag, hipStream_t st);
// mixed precision hand-off: dst = src / ||r0||  (complex64) and X += ||r0|| * D
void fh_launch_narrow_scaled(const cplx* src, size_t src_stride, cplxf* dst, size_t dst_stride, const double* r0norm,
                             int N, int ld, int nblk, int nodes, hipStream_t st);
void fh_launch_widen_axpy(cplx* X, size_t x_stride, const cplxf* D, size_t d_stride, const double* r0norm,
                          int N, int ld, int nblk, int nodes, hipStream_t st);

// ---- device-resident restarted GMRES(m) (fh_gmres.hip) -----------------------------------------
struct fh_gmres_args {
    int N; int mr;                    // restart length
    size_t panel;                     // N * ld elements
    cplx* V; size_t v_node_stride;    // basis panels V[node][0..mr], (mr + 1) * panel apart
    cplx* W;                          // [node] work panel (residual at cycle start, then S v_k)
    cplx* partial;                    // multi-dot partials [node][chunk][blk][8][ld]
    cplx* npartial;                   // norm partials [node][blk][ld]
    cplx* H;                          // [node*ld][(mr+1) x mr] Hessenberg columns (rotated in place)
    cplx* hcur;                       // [node*ld][mr+1] coefficients of the running Gram-Schmidt pass
    cplx* cs; cplx* sn;               // [node*ld][mr] Givens rotations
    cplx* g;                          // [node*ld][mr+1] rotated right-hand side
    cplx* y;                          // [node*ld][mr] solution of the triangular system
    double* inv;                      // [node*ld] column scale of the next basis vector (0 = column finished)
    double* r0norm; double* target; double* rnorm;
    int* active; int* iters; int* status; int* kdim;   // [node*ld]
    int* node_active;                 // [nodes]
};
int fh_gm_nchunk(int k);
size_t fh_gm_partial_elems(int mr, int nblk, int nodes, int ld);
void fh_launch_gm_orthogonalize(const fh_gmres_args& a, int ld, int k, int nblk, int nodes, hipStream_t st);
void fh_launch_gm_start(const fh_gmres_args& a, int ld, int nblk_norm, int nodes, int first, double rtol, double atol, int m, hipStream_t st);
void fh_launch_gm_givens(const fh_gmres_args& a, int ld, int k, int nblk, int nodes, hipStream_t st);
void fh_launch_gm_scale_store(const fh_gmres_args& a, int ld, const cplx* src, size_t src_node_stride, int dst_index, int nblk, int nodes,
                              hipStream_t st);
void fh_launch_gm_finish_cycle(const fh_gmres_args& a, int ld, cplx* X, size_t x_node_stride, int kmax, int nblk, int nodes, hipStream_t st);

// ---- block (panel) operations ------------------------------------------------------------
// column-major (N x m, leading dim lds) <-> row-major panel (N x ld), zero padded
// perm != null: panel row i holds the caller's row perm[i] (block order of a renumbered sparse matrix)
void fh_launch_to_panel(const cplx* src, int64_t lds, int N, int m, cplx* dst, int ld, hipStream_t st, const int* perm = nullptr);
void fh_launch_from_panel(const cplx* src, int ld, int N, int m, cplx* dst, int64_t ldd, hipStream_t st, const int* perm = nullptr);
// resident refinement loop: dst (N x ldd panel, zero padded) = columns [c0, c0 + w) of the N x lds panel src; and the same
// columns written into columns [c0, c0 + w) of an N x ldd array of reals / interleaved complex values (zeros elsewhere)
void fh_launch_panel_cols(const cplx* src, int lds, int c0, int w, int N, cplx* dst, int ldd, hipStream_t st);
void fh_launch_pack_cols(const cplx* src, int lds, int c0, int w, int N, double* dst, int ldd, int real_only, hipStream_t st);
// real column-major source -> complex panel
void fh_launch_to_panel_real(const double* src, int64_t lds, int N, int m, cplx* dst, int ld, hipStream_t st);
// dst = sum_e w[e] * X[e]
void fh_launch_accumulate(const cplx* X, size_t node_stride, const cplx* w, int nodes, int N, int ld, const cplx* extra,
                          cplx* dst, int real_part, hipStream_t st);
// G (ld x ld, column-major, ldg = ld) = X^H Y (bilinear=0) or X^T Y (bilinear=1); f64 MFMA.
// work: at least fh_gram_work_elems(ld) cplx.
size_t fh_gram_work_elems(int ld);
void fh_launch_gram(const cplx* X, const cplx* Y, int N, int ld, int bilinear, cplx* work, cplx* G,
                    hipStream_t st);
// per-column dots: out[c] = <U[:,c], V[:,c]>; work: nblk*ld cplx
int fh_vec_nblk(int N, int ld);
int fh_kry_nblk(int N, int ld, int nodes);
void fh_launch_dot_cols(const cplx* U, const cplx* V, int N, int ld, cplx* work, cplx* out, hipStream_t st);
// X[:,c] *= s[c]
void fh_launch_scale_cols(cplx* X, const cplx* s, int N, int ld, hipStream_t st);
// X[:, c] /= sqrt(dots[c].x) for c < M, on the device (dots: per-column <x, x> from fh_launch_dot_cols)
void fh_launch_normalize_cols(cplx* X, const cplx* dots, int N, int ld, int M, hipStream_t st);
// Xout = Q * V   (V: ld x ld column-major on device, zero padded)
void fh_launch_small_matmul(const cplx* Q, const cplx* V, int N, int ld, cplx* Xout, hipStream_t st);
// dst[:,k] = src[:,perm[k]] for k < count else 0
void fh_launch_gather_cols(const cplx* src, const int* perm, int count, int N, int ld, cplx* dst, hipStream_t st);

// column-pivoted modified Gram-Schmidt with re-orthogonalisation, fully on device.
// state: int[4 + ld] = {k (steps done), rank, done, pivot, perm[ld]}; dstate: double[2+ld] = {R11, thr, norm2[ld]}
struct fh_mgs_args {
    cplx* X; int N; int ld; int m;
    int* istate; double* dstate; cplx* coef;   // coef[ld]
    cplx* work;                                // nblk*ld partials
    double rank_tol;
    double ref_scale = 0.0;                    // external R_11 (block of a wider matrix), 0 = own first pivot
    int big_dim = 0;                           // total column count of the wider matrix (eps*max(N,M0) term)
};
void fh_mgs_run(const fh_mgs_args& a, hipStream_t st);

// dense kernels ---------------------------------------------------------------------------
struct fh_dense;
