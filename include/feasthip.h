/*
 * feasthip.h -- C ABI of libfeasthip.so: the MI355X (gfx950) FEAST contour-integration
 * inner loop that drops in behind FeastKit.jl's feast()/feast_general()/pfeast_* and RCI
 * surfaces as a new `:hip` backend.
 *
 * The reference (subhk/FeastKit.jl v1.0.11) is pure Julia with no FFI of its own; the
 * boundary below is derived from the seams an accelerator can sit behind without
 * touching src/core or src/interfaces.  Each entry point cites
 * the reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - POD arguments only.  No callbacks, no ownership transfer, no exceptions.
 *   - Host matrices are COLUMN-MAJOR (Julia/Fortran); complex data is interleaved
 *     (re,im) double pairs (Julia ComplexF64 / C99 double _Complex layout).
 *   - "_dev" variants take DEVICE pointers in the same column-major layout and run
 *     asynchronously on the handle's stream; the plain variants take host pointers,
 *     copy in/out synchronously and never retain the pointer after return.
 *   - Return value: 0 on success, else the reference's FeastError codes
 *     (src/core/feast_types.jl:257-268): 1 N, 2 M0, 3 Emin/Emax, 4 Emid/r,
 *     5 no convergence, 6 memory, 7 internal (HIP runtime), 8 LAPACK/singular, 9 fpm.
 *     feasthip_last_error() returns a human-readable string for the last failure.
 *   - One handle = one GPU = one host thread at a time (the reference's :threads
 *     backend must not share a handle across Julia threads).
 */
#ifndef FEASTHIP_H
#define FEASTHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEASTHIP_VERSION_MAJOR 0
#define FEASTHIP_VERSION_MINOR 1

/* FeastError, src/core/feast_types.jl:257-268 */
enum {
    FEASTHIP_SUCCESS = 0,
    FEASTHIP_ERROR_N = 1,
    FEASTHIP_ERROR_M0 = 2,
    FEASTHIP_ERROR_EMIN_EMAX = 3,
    FEASTHIP_ERROR_EMID_R = 4,
    FEASTHIP_ERROR_NO_CONVERGENCE = 5,
    FEASTHIP_ERROR_MEMORY = 6,
    FEASTHIP_ERROR_INTERNAL = 7,
    FEASTHIP_ERROR_LAPACK = 8,
    FEASTHIP_ERROR_FPM = 9
};

/* shifted-system solver kinds: keyword `solver` of feast_sygv!/feast_scsrgv!
 * (src/dense/feast_dense.jl:81-84, src/sparse/feast_sparse.jl:249-252).          */
enum {
    FEASTHIP_SOLVER_LU = 0,        /* :direct  -- dense only (batched complex LU)          */
    FEASTHIP_SOLVER_BICGSTAB = 1,  /* :iterative -- batched BiCGStab (Krylov.bicgstab,
                                      src/interfaces/feast_matfree.jl:716)                 */
    FEASTHIP_SOLVER_GMRES = 2,     /* :gmres   -- batched restarted GMRES(m)
                                      (src/sparse/feast_sparse.jl:183-188)                 */
    FEASTHIP_SOLVER_BANDED = 4,    /* sparse DIRECT solver for CSR input: one LU of z B - A per quadrature node, factors
                                      cached across refinement loops.  The direct solver of the banded drivers
                                      (src/banded/feast_banded.jl:100-150) and the build's counterpart of the sparse
                                      drivers' default `lu(z B - A)` (UMFPACK, src/sparse/feast_sparse.jl:339-342).
                                      A narrow band is eliminated as stored (ZGBTRF/ZGBTRS semantics); any other
                                      pattern by a multifrontal LU on a nested-dissection tree (batched dense fronts on
                                      the MFMA LU kernels, partial pivoting inside the fully-summed blocks) when that is
                                      less than half the work of the alternative: reverse Cuthill-McKee + a blocked band
                                      LU on the same kernels (fill confined to the band, 16 N (2 kl + ku + 256) bytes
                                      per node).  feasthip_band_plan / feasthip_direct_plan_flops say which, and what
                                      it costs                                                               */
    FEASTHIP_SOLVER_COCG = 3       /* conjugate-orthogonal CG for the complex-SYMMETRIC shifted
                                      systems that real-symmetric A, B produce (one operator
                                      application per iteration); not in the reference      */
};

enum { FEASTHIP_STORAGE_CSR = 0, FEASTHIP_STORAGE_CSC = 1 };

typedef struct feasthip_ctx* feasthip_handle;

/* Per-call statistics of feasthip_contour_apply (no reference counterpart; replaces the
 * @warn/@debug diagnostics of src/parallel/feast_parallel.jl:266-273).                  */
typedef struct feasthip_stats {
    double  seconds_total;       /* wall time of the call (host clock)                       */
    double  seconds_solve;       /* device time inside shifted solves (HIP events)            */
    int64_t krylov_iterations;   /* sum over local nodes of max-over-columns iterations       */
    int64_t spmm_calls;          /* operator applications (block of m columns each)           */
    int64_t factorizations;      /* dense LU factorizations performed in this call            */
    double  max_rel_residual;    /* max over nodes/columns of ||b - S y|| / ||b|| (iterative) */
} feasthip_stats;

/* ---- lifecycle ------------------------------------------------------------------- */
int  feasthip_version(int* major, int* minor);
/* device_id: HIP ordinal (one process per GPU: LOCAL_RANK). */
int  feasthip_create(feasthip_handle* out, int device_id);
int  feasthip_destroy(feasthip_handle h);
const char* feasthip_last_error(feasthip_handle h);
/* Run all work on an externally owned hipStream_t (e.g. torch's current stream). NULL
 * restores the handle's own stream. */
int  feasthip_set_stream(feasthip_handle h, void* hip_stream);
int  feasthip_synchronize(feasthip_handle h);

/* ---- multi-GPU: one handle per GPU, one process (or task) per handle ---------------------------- */
/* The reference reduces inside its parallel backends: MPI.Allreduce(local_Aq/local_Sq/local_Q, +, comm)
 * (src/parallel/feast_mpi.jl:117-119, 856-858, 1001) and the master sum over worker results
 * (src/parallel/feast_parallel.jl:497-503).  Here the handle owns the collective: once a communicator is
 * attached, feasthip_contour_apply[_dev] returns Qproj (and zAq/zSq, node_status) already summed over all
 * ranks -- ONE packed RCCL all-reduce over xGMI per refinement loop, ordered on the handle's stream.
 *
 * feasthip_comm_unique_id: rank 0 calls it (ncclGetUniqueId) and ships the 128 bytes to the other ranks by
 *   whatever the host has (MPI.bcast in a Julia host, a TCP store in Python).
 * feasthip_comm_init_rank: collective; attaches rank `rank` of `nranks` to the handle's device.
 *   transport FEASTHIP_COMM_RCCL: librccl (resolved at run time), one rank per GPU.
 *   transport FEASTHIP_COMM_SHM : for ranks that SHARE a HIP device (RCCL refuses that); rendezvous through
 *     POSIX shared memory, peers' staging buffers mapped with HIP IPC, summed in rank order by a kernel
 *     (bitwise identical on every rank).  A rehearsal transport for one-GPU test rigs, host-synchronous.
 *   transport FEASTHIP_COMM_AUTO: RCCL unless the environment says FEASTHIP_COMM_TRANSPORT=shm.
 * feasthip_allreduce_sum_dev: in-place SUM of `count` doubles at device pointer dptr over the ranks (the
 *   reduction contour_apply uses, exposed for the host's own small reductions, e.g. residual blocks
 *   src/parallel/feast_mpi.jl:264-284).  Returns after the result is visible.                          */
#define FEASTHIP_UNIQUE_ID_BYTES 128
enum { FEASTHIP_COMM_AUTO = 0, FEASTHIP_COMM_RCCL = 1, FEASTHIP_COMM_SHM = 2 };
int  feasthip_comm_unique_id(char* uid /* FEASTHIP_UNIQUE_ID_BYTES */);
int  feasthip_comm_init_rank(feasthip_handle h, int nranks, int rank, const char* uid, int transport);
int  feasthip_comm_destroy(feasthip_handle h);
int  feasthip_comm_info(feasthip_handle h, int* nranks, int* rank, int* transport);
int  feasthip_allreduce_sum_dev(feasthip_handle h, void* dptr, int64_t count);

/* ---- problem definition ---------------------------------------------------------- */
/* Dense A (and B, NULL => identity), column-major, lda/ldb >= N.
 * Replaces the Matrix arguments of feast_sygv!/feast_hegv!/feast_gegv!
 * (src/dense/feast_dense.jl:356, 402).  is_complex: 0 => double, 1 => interleaved c128. */
int  feasthip_set_dense(feasthip_handle h, int64_t N, int is_complex,
                        const void* A, int64_t lda, const void* B, int64_t ldb);

/* Sparse A (and B, ptrB==NULL => identity).  Julia passes SparseMatrixCSC{T,Int64}
 * (index_base 1, storage CSC); scipy passes CSR base 0.  For CSC input the library
 * transposes on ingest, so results are for the matrix as the CALLER defines it
 * (no silent conj/transpose).
 * Replaces the SparseMatrixCSC arguments of feast_scsrgv!/feast_hcsrgv!/feast_gcsrgv!
 * (src/sparse/feast_sparse.jl:713, 873) and pfeast_scsrgv! (src/parallel/feast_parallel.jl:450). */
int  feasthip_set_csr(feasthip_handle h, int64_t N, int is_complex, int index_base, int storage,
                      int64_t nnzA, const int64_t* ptrA, const int64_t* idxA, const void* valA,
                      int64_t nnzB, const int64_t* ptrB, const int64_t* idxB, const void* valB);

/* Contour nodes/weights as produced by feast_contour / feast_gcontour
 * (src/core/feast_tools.jl:212-371): zne/wne are 2*ne doubles (re,im interleaved).
 * weight_scale = 2.0 for the Hermitian half contour (src/dense/feast_dense.jl:174),
 * 1.0 for the general full contour (src/kernel/feast_kernel.jl:762-766).              */
int  feasthip_set_contour(feasthip_handle h, int ne, const double* zne, const double* wne,
                          double weight_scale);

/* real_part = 1: Qproj (and zAq/zSq) receive only the REAL part of the weighted sum,
 * Qproj = Re( sum_e 2 w_e Y_e ).  For real-symmetric A, B and a real Q this equals the sum
 * over the full contour (the conjugate half is the complex conjugate), i.e. the true FEAST
 * rational filter -- what the reference's real paths do: _pfeast_store_real_moments!
 * (src/parallel/feast_parallel.jl:38-55), feast_srci! (src/kernel/feast_kernel.jl:143,183-186).
 * real_part = 0 (default) keeps the complex half-contour sum of the complexified variant A
 * (src/dense/feast_dense.jl:231).                                                        */
int  feasthip_set_real_projection(feasthip_handle h, int real_part);

/* Restrict this handle to nodes [first, first+count) -- the block partition of
 * distribute_contour_points (src/parallel/feast_parallel.jl:433-447) /
 * MPIFeastState (src/parallel/feast_mpi.jl:36-43).  Default: all nodes.              */
int  feasthip_set_node_range(feasthip_handle h, int first, int count);
/* Arbitrary node subset (0-based contour indices).  Used by the iterative solvers to pair
 * near-axis (slow) with far-axis (fast) nodes per GPU instead of contiguous blocks; the
 * summed result is independent of the assignment.                                       */
int  feasthip_set_node_list(feasthip_handle h, int count, const int* indices);

/* Column block of this handle: the following contour_apply calls sweep only columns [first, first+count) of the
 * m columns they are given (the rest of Qproj is zero on this rank and filled in by the all-reduce); count < 0
 * restores "all columns".  The reference shards quadrature nodes only (src/parallel/feast_parallel.jl:433-447);
 * with Krylov solves the iteration counts of the nodes differ by 10x while the columns of a node cost the same,
 * so ranks are arranged as (node groups) x (column groups).                                                  */
int  feasthip_set_column_block(feasthip_handle h, int64_t first, int64_t count);

/* Inexact-FEAST extension (not in the reference): columns c < m with mask[c] == 0 keep their initial
 * guess (the Ritz warm start q_c/(z - lambda_c) when ritz_lambda is given) and are never iterated by
 * the BiCGStab / COCG solvers of the following contour_apply calls (the restarted GMRES path ignores the
 * mask: all columns of a node advance in lock-step through one Arnoldi basis).  Used to stop spending solves on the
 * guard columns (Ritz values outside the interval) once the subspace has settled.  mask == NULL or
 * m == 0 clears it.  Ignored by the LU path.  The mask is ONE-SHOT: it is consumed by the next
 * contour_apply call (cleared when that call returns, whatever its outcome) and never applies to
 * feasthip_shifted_solve.                                                                    */
int  feasthip_set_column_mask(feasthip_handle h, int64_t m, const int* mask);

/* Error 7 with "handle poisoned" in feasthip_last_error: a Krylov sweep missed its progress deadline or the device
 * queue faulted, and kernels of the failed call may still be in flight.  Every later call on the handle fails fast;
 * feasthip_destroy then neither waits for the stream nor frees the workspaces.  The host must end the process with a
 * non-zero status (or continue in a fresh child process) -- never re-exec a process that has touched the GPU.  */

/* Solver options: keyword args solver/solver_tol/solver_maxiter/solver_restart
 * (src/dense/feast_dense.jl:81-84).  Iterative stop test is Krylov.jl's
 * ||r_k|| <= atol + rtol*||r_0|| per column.  factor_precision 64|32: 32 = mixed precision.  Dense LU:
 * complex64 factors and substitutions (f32 MFMA) inside an fp64 iterative-refinement loop (residual by the
 * fp64 operator kernel, stop at 1e-14 relative, <= 8 steps) -- fp64-accurate solves while
 * cond(zB-A)*eps32 < 1 (the reference has no counterpart; BASELINE config 5 asks for it).
 * BiCGStab/COCG: correction solve on complex64 panels around an fp64 residual (inexact-solve mode).
 * cache_factors: keep LU factors per
 * node across calls (src/dense/feast_dense.jl:147,188).                                 */
/* Free the cached factorisations of the direct solvers (dense LU, band LU): the reference keeps `lu(z B - A)` per node for
 * the life of a driver call (factor cache, src/dense/feast_dense.jl:458, 487-497) and the garbage collector takes them;
 * here they live until the problem changes, the handle is destroyed, or this call.  The next direct solve factors again.  */
int  feasthip_release_factors(feasthip_handle h);

/* The band FEASTHIP_SOLVER_BANDED would eliminate for the current CSR problem (after its reordering) and the device memory
 * of ONE node's factor; *blocked = 1 when the blocked band LU on the dense kernels is used (2: multifrontal, below).  A host shim uses it to decide
 * between the direct and the iterative solvers (the reference decides by keyword only: src/sparse/feast_sparse.jl:249-252).
 * Returns 0, FEASTHIP_ERROR_FPM when no CSR problem is set or the band is beyond the solver's reach.                     */
int  feasthip_band_plan(feasthip_handle h, int* kl, int* ku, int64_t* bytes_per_node, int* blocked);
/* Real flops of ONE node's factorisation under the plan feasthip_band_plan reports (*blocked = 2: the multifrontal
 * elimination on a nested-dissection tree, the counterpart of the reference's UMFPACK call src/sparse/feast_sparse.jl:334-342,
 * taken when its work is under half the band elimination's; kl / ku are then the band it replaced and bytes_per_node the
 * multifrontal factors).                                                                                                 */
int  feasthip_direct_plan_flops(feasthip_handle h, double* flops_per_node);

int  feasthip_set_solver(feasthip_handle h, int kind, double rtol, double atol, int maxit,
                         int restart, int factor_precision, int cache_factors);

/* ---- host policy of the inexact FEAST mode (no device work; needs no problem and no GPU) -----------------------------
 * What the :hip backend adds to the reference's driver loop when `solver = :direct` on large sparse input is served by the
 * warm-started, inexact Krylov sweeps (not in the reference; DESIGN.md sections 2 and 5): which ellipse ratio fpm[18] to put
 * the Gauss / trapezoid nodes on, how many inner iterations and which inner tolerance the next sweep gets, and which Ritz
 * pairs are solver noise.  Kept here so that a host shim CALLS it instead of re-porting it:
 *   feasthip_policy_init     state for one solve.  steer != 0: the library picks fpm[18] itself (the caller left it unset),
 *                            p->aspect holds the a-priori choice; steer == 0: p->aspect = fpm18 stays, only the guards act.
 *   feasthip_policy_update   after every refinement loop that did not converge: epsout, the count M of Ritz values inside,
 *                            whether a node stopped at the iteration cap (node_status 5), the rank Ritz values ordered
 *                            inside-first.  On return p->aspect (changed: recompute the contour with it, feasthip_set_contour),
 *                            p->inner_cap and p->next_rtol (feasthip_set_solver) describe the NEXT sweep.
 *   feasthip_policy_set_aside  flags[j] = 1 for pairs inside the interval that are solver noise (residual > 0.1 and > 100 x
 *                            the smallest residual inside); returns their number (0 when all M would be flagged).
 *   feasthip_policy_filter_ratio / _reach: the filter model the steering rests on (for tests and diagnostics).        */
typedef struct feasthip_policy {
    double Emin, Emax, inner_rtol, outer_tol;
    int ne, quadrature;          /* fpm[2], fpm[16] (0 Gauss, 1 trapezoid)                                              */
    int steer;                   /* 1: the policy picks fpm[18]; 0: guards only                                          */
    int aspect, cap;             /* fpm[18] of the next sweep; upper bound the safeguard has put on it (8000 at first)  */
    int inner_cap, base_cap;     /* Krylov iteration cap per loop: next sweep / as given                                */
    int n_hist;
    double eps_prev;             /* outer residual of the previous loop (inf before the first)                          */
    double eps_hist[3];          /* stagnation guard                                                                    */
    double next_rtol;            /* inner relative tolerance of the next sweep                                          */
    double last_reach;           /* subspace reach the last update steered by (< 0: none)                               */
} feasthip_policy;
int    feasthip_policy_init(feasthip_policy* p, double Emin, double Emax, int ne, int quadrature, double inner_rtol,
                            double outer_tol, int solver_maxiter, int steer, int fpm18);
int    feasthip_policy_update(feasthip_policy* p, double epsout, int M, int any_node_capped, const double* ritz, int nritz);
int    feasthip_policy_set_aside(const double* res, int M, int* flags);
double feasthip_policy_filter_ratio(double Emin, double Emax, int ne, int quadrature, int fpm18, double reach,
                                    const double* inside, int n_inside);
double feasthip_policy_reach(const double* ritz, int n, double Emin, double Emax, double quantile);

/* ---- the hot path ------------------------------------------------------------------ */
/* One contour sweep over this handle's node range:
 *     for e in local nodes:  Y_e = (z_e B - A)^{-1} (B Q);   Qproj += weight_scale*w_e*Y_e
 *     optionally  zAq += weight_scale*w_e * Q^H Y_e,  zSq += weight_scale*w_e*z_e * Q^H Y_e
 * Q, Qproj: N x m c128 column-major (ldq = N), 1 <= m <= N; m > 64 is processed in 64-column
 * panels (this holds for every entry point below; zAq/zSq need m <= 64).  Qproj is OVERWRITTEN with this handle's
 * partial sum (callers reduce across handles/ranks: src/parallel/feast_parallel.jl:497-503,
 * src/parallel/feast_mpi.jl:117-119).  zAq/zSq: m x m c128 or NULL.
 * ritz_lambda: NULL => zero initial guess (reference behaviour, Krylov.jl gmres);
 *   else m doubles (real Ritz values paired with the columns of Q, as in
 *   Q_basis <- solutions of src/dense/feast_dense.jl:336-337): iterative solvers start
 *   from Y0[:,j] = Q[:,j]/(z_e - lambda_j).  Ignored by the LU solver.
 * node_status[e_local]: 0 ok, 5 not converged, 8 singular.  With a communicator attached node_status is
 *   GLOBAL: ne entries indexed by contour node (every rank receives the same vector).
 * Replaces: loop bodies src/dense/feast_dense.jl:171-232, src/sparse/feast_sparse.jl:318-370,
 * workers pfeast_solve_sparse_single_point (src/parallel/feast_parallel.jl:717-751) and
 * mpi_compute_local_moments (src/parallel/feast_mpi.jl:206-253).                          */
int  feasthip_contour_apply(feasthip_handle h, int64_t m, const void* Q, const double* ritz_lambda,
                            void* Qproj, void* zAq, void* zSq, int* node_status,
                            feasthip_stats* stats);
int  feasthip_contour_apply_dev(feasthip_handle h, int64_t m, const void* dQ,
                                const double* ritz_lambda_host, void* dQproj,
                                void* dzAq, void* dzSq, int* node_status, feasthip_stats* stats);

/* ---- the refinement loop with resident panels ---------------------------------------------------------------------
 * One loop of variant A -- contour sweep, _feast_qr_compress!, the reduced pencil, Ritz vectors, residuals
 * (src/dense/feast_dense.jl:171-337, src/sparse/feast_sparse.jl:318-478) -- as three calls whose N x m blocks never
 * leave the device NOR the kernels' own layout between them (m <= 64, no moment matrices).  The per-primitive entry
 * points below compute the same quantities through column-major blocks at every call; a shim that does not need the
 * intermediate blocks on the host should prefer these.
 *
 * feasthip_contour_apply_resident: the sweep of feasthip_contour_apply_dev.  dQ != NULL: the N x m subspace (column
 *   major, device) is imported; dQ == NULL: the Ritz vectors left by the last feasthip_rr_ritz_resident are the subspace
 *   (m must be that call's r) and, when ritz_lambda equals that call's lambda, its residual panel A X - B X diag(lambda)
 *   starts the warm-started Krylov sweeps (one operator product saved).  Q_proj stays resident, summed over the ranks of
 *   an attached communicator (node_status as in feasthip_contour_apply_dev).
 * feasthip_rr_reduce_resident: *rank = numerical rank of the resident Q_proj under the reference's rule
 *   (src/core/feast_aux.jl:101-131); Aq and Bq receive the rank x rank pencil (Q_b^H A Q_b, Q_b^H B Q_b) (column-major, host;
 *   Hermitian parts when hermitize != 0, src/core/feast_aux.jl:84-92) of a basis Q_b of its range, to be handed to the
 *   generalized reduced eigensolver as the reference does (eigen(Hermitian(Sq), Hermitian(Aq)), src/dense/feast_dense.jl:272).
 *   Q_b is Q_proj with its columns scaled to unit length when that basis is well conditioned (the steady state of FEAST:
 *   the test is the one-pass condition of the Cholesky-QR, pivot ratio of the equilibrated Gram matrix > 1e-2) -- no
 *   orthonormal basis is formed, Bq is then the equilibrated Gram-type matrix, not I; otherwise Q_b is the orthonormal
 *   basis of the rank-revealing orthonormalisation (Bq = exactly I for B = I, src/dense/feast_dense.jl:255-259).
 * feasthip_rr_ritz_resident: X = Q_b V for the r = rank columns of V (r x r, host, column-major), the first M columns
 *   normalised when normalize != 0, res[j] = ||A x_j - lambda_j B x_j|| / max(|lambda_j|, 1) for j < M (use_B = 0: without
 *   B, the RCI kernels' residual); lambda: r complex values (2 r doubles).  X stays resident as the next sweep's subspace.
 *   May be called again with another V / M for the same reduction (spurious-pair reordering).
 * feasthip_resident_export: column-major copy (N x ncols, device) of the first ncols resident Ritz vectors (which = 0) or
 *   of the resident Q_proj (which = 1): the converged eigenvectors leave the device once per solve.
 * Replaces the bodies of src/dense/feast_dense.jl:234-337 and src/sparse/feast_sparse.jl:372-478 between two sweeps.      */
int  feasthip_contour_apply_resident(feasthip_handle h, int64_t m, const void* dQ, const double* ritz_lambda_host,
                                     int* node_status, feasthip_stats* stats);
int  feasthip_rr_reduce_resident(feasthip_handle h, int64_t m, double rank_tol, int hermitize, int* rank,
                                 void* Aq_host, void* Bq_host);
int  feasthip_rr_ritz_resident(feasthip_handle h, int64_t r, const void* V_host, const double* lambda_host, int64_t M,
                               int normalize, int use_B, double* res_host);
int  feasthip_resident_export(feasthip_handle h, int which, int64_t ncols, void* dX);
/* The counterpart: a column-major N x ncols device block becomes the resident subspace (which = 0: resume from a saved
 * subspace, the reference's fpm[5] = 1 start) or the resident Q_proj (which = 1: a projection computed elsewhere). */
int  feasthip_resident_import(feasthip_handle h, int which, int64_t ncols, const void* dX);

/* Rank-revealing orthonormalisation of Q[:, 0:m] in place.  Column-pivoted
 * Gram-Schmidt with re-orthogonalisation; rank = #{ |R_ii| > max(rank_tol, eps*max(N,m))*|R_11| },
 * the rule of _feast_qr_compress! (src/core/feast_aux.jl:101-131).  On return the first
 * `rank` columns of Q hold the orthonormal basis.                                         */
int  feasthip_orthonormalize(feasthip_handle h, int64_t m, void* Q, double rank_tol, int* rank);
int  feasthip_orthonormalize_dev(feasthip_handle h, int64_t m, void* dQ, double rank_tol, int* rank);

/* Rayleigh-Ritz projection: Aq = herm(Q^H A Q), Bq = herm(Q^H B Q) (Bq = I when
 * B is the identity).  bilinear=1 uses Q^T (complex-symmetric siblings) and skips the
 * Hermitian symmetrisation.  hermitize=0 returns the raw products, Bq = Q^H Q for B = I (variant C,
 * src/kernel/feast_kernel.jl:790,805).  Replaces src/dense/feast_dense.jl:252-265,
 * src/sparse/feast_sparse.jl:392-405.  Aq, Bq: r x r c128 column-major HOST buffers.       */
int  feasthip_project(feasthip_handle h, int64_t r, const void* Q, int bilinear, int hermitize,
                      void* Aq, void* Bq);
int  feasthip_project_dev(feasthip_handle h, int64_t r, const void* dQ, int bilinear, int hermitize,
                          void* Aq_host, void* Bq_host);

/* Ritz back-transform + residuals (SURVEY a12,a13):  X = Q*V (N x r), optional column
 * normalisation of the first M columns, then res_j = ||A x_j - lambda_j B x_j||_2 / max(|lambda_j|,1)
 * for j < M (use_B=0 drops B as the RCI kernels do, src/kernel/feast_kernel.jl:899-906).
 * V: r x r c128 column-major host; lambda: r c128 host (re,im).
 * Replaces src/dense/feast_dense.jl:287-322, src/core/feast_tools.jl:726-755.              */
int  feasthip_ritz_residual(feasthip_handle h, int64_t r, const void* Q, const void* V,
                            const double* lambda, int64_t M, int normalize, int use_B,
                            void* X, double* res);
int  feasthip_ritz_residual_dev(feasthip_handle h, int64_t r, const void* dQ, const void* V_host,
                                const double* lambda_host, int64_t M, int normalize, int use_B,
                                void* dX, double* res_host);

/* ---- RCI / matrix-free seams -------------------------------------- */
/* Rayleigh-Ritz step with the reduced eigenproblem on the device:
 * feasthip_project, then the Hermitian-definite r x r pencil is solved by a Jacobi eigensolver in LDS
 * (instead of eigen(Hermitian(Sq), Hermitian(Aq)), src/dense/feast_dense.jl:272), stable inside-first
 * reorder for [Emin, Emax] (src/core/feast_aux.jl:144-197), then feasthip_ritz_residual.  r <= 64.
 * Out: X (N x r device), lambda[r] (reordered, inside first), *M, res[M].  Returns 8 when the reduced
 * B matrix is not positive definite (use project + a host general eigensolver + ritz_residual then,
 * the fallback of src/dense/feast_dense.jl:276-284).                                            */
int  feasthip_rayleigh_ritz_dev(feasthip_handle h, int64_t r, const void* dQ, double Emin, double Emax,
                                int use_B, void* dX, double* lambda_out, int* M_out, double* res_out);

/* Y = op * X for op = A (which=0) or B (which=1): RCI jobs 30 / 40
 * (src/core/feast_types.jl:227-249; callers src/dense/feast_dense.jl:561-573).          */
int  feasthip_matmul(feasthip_handle h, int which, int64_t m, const void* X, void* Y);
int  feasthip_matmul_dev(feasthip_handle h, int which, int64_t m, const void* dX, void* dY);

/* Y = (z B - A)^{-1} X, all m columns: RCI jobs 10+11 and the matrix-free
 * linear_solver(Y, z, X) callback contract (src/interfaces/feast_matfree.jl:149, 697).
 * Unlike contour_apply the right-hand side is X itself (the caller applies B).          */
int  feasthip_shifted_solve(feasthip_handle h, double z_re, double z_im, int64_t m,
                            const void* X, void* Y, feasthip_stats* stats);
int  feasthip_shifted_solve_dev(feasthip_handle h, double z_re, double z_im, int64_t m,
                                const void* dX, void* dY, feasthip_stats* stats);

/* ---- measurement support ---------------------------------------------------------- */
/* Average device time (ms, HIP events on the handle's stream) and launch count of the
 * named kernel class since the last reset; classes: "spmm", "bicg_update", "dot_finalize",
 * "lu_panel", "lu_gemm", "trsm", "gram", "ortho".  Used by bench.py's roofline object.     */
/* Per local node: iterations of the slowest column in the last iterative sweep. */
int  feasthip_last_node_iterations(feasthip_handle h, int* out, int n);
/* With a communicator: Krylov iterations of the last sweep per CONTOUR node (ne entries), summed over the ranks that
 * worked on the node (they travel in the tail of the packed all-reduce).  Identical on every rank, so hosts can
 * re-balance the node lists between refinement loops deterministically (nodes next to the real axis need 10x the
 * iterations of the others).                                                                              */
int  feasthip_last_global_node_iterations(feasthip_handle h, int* out, int n);
/* [local node][m] iterations per column of the last iterative sweep (row-major, n entries). */
int  feasthip_last_column_iterations(feasthip_handle h, int* out, int n);
int  feasthip_profile_enable(feasthip_handle h, int enable);
int  feasthip_profile_reset(feasthip_handle h);
int  feasthip_profile_get(feasthip_handle h, const char* kernel_class, double* total_ms,
                          int64_t* launches);
/* Sampling period of the event timing: 0 = default (1 launch in 13), 1 = every launch (classes with few, very
 * different launches: the dense LU).  feasthip_profile_get_work: algorithmic flops issued by a dense MFMA class
 * ("lu_gemm", "lu_gemm_in") since the last reset, for the MFMA roofline of bench.py.                        */
int  feasthip_profile_set_period(feasthip_handle h, int period);
int  feasthip_profile_get_work(feasthip_handle h, const char* kernel_class, double* work);

#ifdef __cplusplus
}
#endif
#endif /* FEASTHIP_H */
