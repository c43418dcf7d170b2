// fh_dense.hpp -- dense shifted systems: operator application on panels and the batched
// complex LU (form z B - A, factor with partial pivoting, solve) for the FEAST contour sweep.
#pragma once
#include "fh_common.hpp"
#include <vector>

struct fh_dense_op_args {
    const void* A; const void* B;   // N x N column-major, double or cplx; B may be null (identity)
    int N; int is_complex; int nodes;
    const cplx* X; size_t x_node_stride;
    cplx* Y; size_t y_node_stride;
    const cplx* coefA; const cplx* coefB;
    const cplx* Bvec; size_t b_node_stride;
    const cplx* U; size_t u_node_stride;
    int dot_mode; cplx* partial1; cplx* partial2;
    const int* node_active;
};
int fh_dense_op_nblk(int N);
void fh_launch_dense_op(const fh_dense_op_args& a, int ld, int nblk, hipStream_t st);

// R -= X diag(lam)
void fh_launch_axpy_cols(cplx* R, const cplx* X, const cplx* lam, int N, int ld, hipStream_t st);

// Factor (cached per local node when h->cache_factors) and solve all local nodes:
//   Y[e] = (z_e B - A)^{-1} RHS     RHS: one shared panel; Y: node-strided panels
int fh_dense_lu_solve_nodes(feasthip_ctx* h, int ld, int m, int nodes, const std::vector<cplx>& z, const cplx* RHS, size_t rhs_stride,
                            cplx* Y, size_t stride, std::vector<int>& status, int64_t* nfact);
// one-off (uncached) solve for a single shift
int fh_dense_lu_solve_single(feasthip_ctx* h, int ld, int m, cplx z, const cplx* RHS, cplx* Y, int* status,
                             int64_t* nfact);

// Blocked band LU on the dense kernels (fh_dense.hip, "wide band"): storage per node fh_wband_elems complex128 values, the
// matrix pointer the kernels take is storage + fh_wband_base_offset.
size_t fh_wband_elems(int N, int kl, int ku);
size_t fh_wband_base_offset(int N, int kl, int ku);
// prec: 64 = complex128 factors, 32 = complex64 factors (storage elements of that type; refinement is the caller's)
int fh_wband_factor(feasthip_ctx* h, int prec, int nf, void* const* abs_host, void** dbases, int** dpvs, const cplx* dz, int* dinfo,
                    const int* d_iperm, int kl, int ku);
int fh_wband_solve(feasthip_ctx* h, int prec, int nf, void** dbases, int** dpvs, int** dperms, const int* d_perm, const cplx* RHS, size_t rhs_stride,
                   cplx* Y, size_t stride, void* Yb, void* Zb, int ld, int m, int kl, int ku);

// Multifrontal sparse LU (fh_dense.hip, symbolic phase fh_mf.hpp): plan from the host pattern, batched numeric factorisation
// of nf shifted matrices into per-node stores, substitution through the elimination tree.  prec 64: complex128 factors; 32: complex64 factors (refinement is the caller's).
int fh_mf_make_plan(feasthip_ctx* h, int leaf);
void fh_mf_free(feasthip_ctx* h);
double fh_mf_plan_flops(feasthip_ctx* h);
int fh_mf_max_front(feasthip_ctx* h);
int fh_mf_max_group(feasthip_ctx* h);
size_t fh_mf_store_bytes(feasthip_ctx* h, int prec);
size_t fh_mf_pivot_ints(feasthip_ctx* h);
size_t fh_mf_work_bytes(feasthip_ctx* h, int prec);
int fh_mf_factor(feasthip_ctx* h, int prec, int nf, void* const* stores, int* const* pivs, const cplx* dz, std::vector<int>& info_out);
int fh_mf_solve(feasthip_ctx* h, int prec, int nf, void* const* stores, int* const* pivs, const cplx* RHS, size_t rhs_stride, cplx* OUT, size_t out_stride, int ld, int m);
