// Reduced Hermitian-definite eigenproblem on the device:
//     S v = lambda A v,   S = Q^H A_op Q (Hermitian),  A = Q^H B_op Q (Hermitian positive definite, or I),
// r <= 64, replacing eigen(Hermitian(Sq), Hermitian(Aq)) of src/dense/feast_dense.jl:272 (ZHEGV).
// One workgroup: Cholesky A = L L^H, C = L^-1 S L^-H, cyclic two-sided Jacobi on C held in LDS
// (round-robin ordering: n/2 disjoint rotations per round, three barriers per round), eigenvalues
// sorted ascending like LAPACK, eigenvectors W = L^-H V so that W^H A W = I.  A real-symmetric
// input keeps real rotations (the phase of a real off-diagonal entry is +-1), so the Ritz vectors of
// the real-projection path stay real.
#include <cstdlib>

#include "fh_eig.hpp"

#define EIG_THREADS 1024
#define EIG_MAX 64

__device__ __forceinline__ cplx eig_conj(cplx a) { return cmake(a.x, -a.y); }

// scratch layout (doubles/cplx in global memory, all tiny): Lg[64*64], Vg[64*64], rot[32*4], flags[4]
// VLDS: the eigenvector accumulator and the rotation parameters live in (dynamic) LDS next to C -- 129 KB
// per workgroup, which gfx950's 160 KB LDS allows; otherwise they stay in global memory (L2) and every one
// of the ~1500 phases pays a memory round trip (2.6 ms instead of a fraction of that at r = 64).
template <bool VLDS>
__global__ __launch_bounds__(EIG_THREADS) void k_herm_eig(int r, int ld, const cplx* __restrict__ S,
                                                          const cplx* __restrict__ A, cplx* Lg, cplx* Vglob, double* rotglob,
                                                          int* flags, double* lambda, cplx* Vout, int max_sweeps) {
    extern __shared__ cplx eig_dyn[];
    __shared__ cplx Cst[VLDS ? 1 : EIG_MAX * EIG_MAX];
    cplx* C = VLDS ? eig_dyn : Cst;                       // column-major, stride 64
    cplx* Vg = VLDS ? eig_dyn + EIG_MAX * EIG_MAX : Vglob;
    double* rot = VLDS ? (double*)(eig_dyn + 2 * EIG_MAX * EIG_MAX) : rotglob;
    const int t = threadIdx.x;
    const int n = (r + 1) & ~1;                           // even size for the round-robin pairing
    // ---- load, Hermitian part
    for (int e = t; e < EIG_MAX * EIG_MAX; e += EIG_THREADS) {
        const int i = e & 63, j = e >> 6;
        cplx v = cmake(0, 0);
        if (i < r && j < r) {
            const cplx a = S[i + (size_t)j * ld], b = S[j + (size_t)i * ld];
            v = cmake(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
        } else if (i == j && i < n) {
            v = cmake(1e300, 0);                          // padding index: decoupled, sorts last
        }
        C[e] = v;
        if (!VLDS || !A) Vg[e] = cmake(i == j ? 1.0 : 0.0, 0.0);
    }
    if (t < 4) flags[t] = 0;
    __syncthreads();
    // Cholesky factor: in LDS mode it borrows the eigenvector region until C = L^-1 S L^-H is formed, is
    // then parked in global memory, and returns to LDS (in C's place) for the back-transformation
    cplx* L = (VLDS && A) ? Vg : Lg;
    // ---- generalized problem: C = L^-1 C L^-H
    if (A) {
        for (int e = t; e < r * r; e += EIG_THREADS) {
            const int i = e % r, j = e / r;
            const cplx a = A[i + (size_t)j * ld], b = A[j + (size_t)i * ld];
            L[i + j * EIG_MAX] = cmake(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
        }
        __syncthreads();
        for (int k = 0; k < r; ++k) {
            if (t == 0) {
                const double d = L[k + k * EIG_MAX].x;
                if (!(d > 0.0) || !isfinite(d)) flags[0] = k + 1;      // not positive definite
                else L[k + k * EIG_MAX] = cmake(sqrt(d), 0);
            }
            __syncthreads();
            if (flags[0]) return;
            const double dk = L[k + k * EIG_MAX].x;
            for (int i = k + 1 + t; i < r; i += EIG_THREADS) L[i + k * EIG_MAX] = cscale(L[i + k * EIG_MAX], 1.0 / dk);
            __syncthreads();
            const int m = r - k - 1;
            for (int e = t; e < m * m; e += EIG_THREADS) {
                const int i = k + 1 + e % m, j = k + 1 + e / m;
                if (j <= i) L[i + j * EIG_MAX] = csub(L[i + j * EIG_MAX], cmul(L[i + k * EIG_MAX], eig_conj(L[j + k * EIG_MAX])));
            }
            __syncthreads();
        }
        if (t < r) {                                       // column t of C <- L^-1 (column t)
            for (int i = 0; i < r; ++i) {
                cplx x = C[i + t * EIG_MAX];
                for (int k = 0; k < i; ++k) x = csub(x, cmul(L[i + k * EIG_MAX], C[k + t * EIG_MAX]));
                C[i + t * EIG_MAX] = cscale(x, 1.0 / L[i + i * EIG_MAX].x);
            }
        }
        __syncthreads();
        if (t < r) {                                       // row t of C <- (row t) L^-H
            for (int j = 0; j < r; ++j) {
                cplx y = C[t + j * EIG_MAX];
                for (int k = 0; k < j; ++k) y = csub(y, cmul(C[t + k * EIG_MAX], eig_conj(L[j + k * EIG_MAX])));
                C[t + j * EIG_MAX] = cscale(y, 1.0 / L[j + j * EIG_MAX].x);
            }
        }
        __syncthreads();
        for (int e = t; e < r * r; e += EIG_THREADS) {     // restore exact Hermitian symmetry
            const int i = e % r, j = e / r;
            if (i < j) {
                const cplx a = C[i + j * EIG_MAX], b = C[j + i * EIG_MAX];
                const cplx v = cmake(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
                C[i + j * EIG_MAX] = v;
                C[j + i * EIG_MAX] = eig_conj(v);
            } else if (i == j) {
                C[i + i * EIG_MAX].y = 0.0;
            }
        }
        __syncthreads();
        if (VLDS) {
            for (int e = t; e < EIG_MAX * EIG_MAX; e += EIG_THREADS) {
                const int i = e & 63, j = e >> 6;
                if (i < r && j < r) Lg[e] = L[e];
                Vg[e] = cmake(i == j ? 1.0 : 0.0, 0.0);
            }
            __syncthreads();
        }
    }
    // ---- cyclic Jacobi, round-robin pairs
    // entries cannot be driven below the rounding level eps*||C||: "still converging" is judged against
    // the larger of the relative (small eigenvalues) and this absolute floor, or the sweeps never stop
    __shared__ double s_scale;
    if (t == 0) {
        double mx = 0.0;
        for (int i = 0; i < r; ++i) mx = fmax(mx, fabs(C[i + i * EIG_MAX].x));
        s_scale = mx;
    }
    __syncthreads();
    const double abs_floor = 4.5e-16 * s_scale;
    const int half = n / 2;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (t == 0) flags[1] = 0;
        __syncthreads();
        for (int round = 0; round < n - 1; ++round) {
            if (t < half) {
                int p, q;
                if (t == 0) { p = n - 1; q = round; }
                else { p = (round + t) % (n - 1); q = (round - t + (n - 1)) % (n - 1); }
                if (p > q) { const int u = p; p = q; q = u; }
                const double app = C[p + p * EIG_MAX].x, aqq = C[q + q * EIG_MAX].x;
                const cplx apq = C[p + q * EIG_MAX];
                const double beta = hypot(apq.x, apq.y);
                double c = 1.0, sx = 0.0, sy = 0.0;
                // skip when the entry is negligible against the diagonal pair (relative criterion)
                if (beta > 1e-300 && beta > 2.3e-16 * 1e-3 * sqrt(fabs(app) * fabs(aqq)) && q < r) {
                    const double tau = (aqq - app) / (2.0 * beta);
                    const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + tt * tt);
                    const double s = tt * c;
                    sx = s * apq.x / beta;                  // s * e^{i phi}
                    sy = s * apq.y / beta;
                    if (beta > fmax(1e-15 * sqrt(fabs(app) * fabs(aqq)), abs_floor)) flags[1] = 1;   // still converging
                }
                rot[4 * t + 0] = c; rot[4 * t + 1] = sx; rot[4 * t + 2] = sy;
                rot[4 * t + 3] = (double)(p * EIG_MAX + q);
            }
            __syncthreads();
            // columns: [col_p, col_q] <- [c col_p - conj(s) col_q,  s col_p + c col_q]   (C and V)
            for (int e = t; e < half * n; e += EIG_THREADS) {
                const int k = e / n, i = e % n;
                const double c = rot[4 * k];
                const cplx s = cmake(rot[4 * k + 1], rot[4 * k + 2]);
                if (s.x == 0.0 && s.y == 0.0) continue;
                const int pq = (int)rot[4 * k + 3], p = pq / EIG_MAX, q = pq % EIG_MAX;
                const cplx cp = C[i + p * EIG_MAX], cq = C[i + q * EIG_MAX];
                C[i + p * EIG_MAX] = csub(cscale(cp, c), cmul(eig_conj(s), cq));
                C[i + q * EIG_MAX] = cadd(cmul(s, cp), cscale(cq, c));
                const cplx vp = Vg[i + p * EIG_MAX], vq = Vg[i + q * EIG_MAX];
                Vg[i + p * EIG_MAX] = csub(cscale(vp, c), cmul(eig_conj(s), vq));
                Vg[i + q * EIG_MAX] = cadd(cmul(s, vp), cscale(vq, c));
            }
            __syncthreads();
            // rows: [row_p; row_q] <- [c row_p - s row_q;  conj(s) row_p + c row_q]
            for (int e = t; e < half * n; e += EIG_THREADS) {
                const int k = e / n, j = e % n;
                const double c = rot[4 * k];
                const cplx s = cmake(rot[4 * k + 1], rot[4 * k + 2]);
                if (s.x == 0.0 && s.y == 0.0) continue;
                const int pq = (int)rot[4 * k + 3], p = pq / EIG_MAX, q = pq % EIG_MAX;
                const cplx rp = C[p + j * EIG_MAX], rq = C[q + j * EIG_MAX];
                cplx np_ = csub(cscale(rp, c), cmul(s, rq));
                cplx nq_ = cadd(cmul(eig_conj(s), rp), cscale(rq, c));
                if (j == p) np_.y = 0.0;                    // diagonal stays real, annihilated entry exact
                if (j == q) nq_.y = 0.0;
                if (j == q) np_ = cmake(0, 0);
                if (j == p) nq_ = cmake(0, 0);
                C[p + j * EIG_MAX] = np_;
                C[q + j * EIG_MAX] = nq_;
            }
            __syncthreads();
        }
        if (t == 0) flags[3] = sweep + 1;
        if (flags[1] == 0) break;                           // uniform: read after the barrier of the last round
        __syncthreads();
    }
    // ---- sort ascending, back-transform, write out
    __shared__ double lam_s[EIG_MAX];
    __shared__ int rank_s[EIG_MAX];
    if (t < r) lam_s[t] = C[t + t * EIG_MAX].x;
    __syncthreads();
    if (t < r) {
        const double li = lam_s[t];
        int rank = 0;
        for (int j = 0; j < r; ++j) {
            const double lj = lam_s[j];
            if (lj < li || (lj == li && j < t)) ++rank;
        }
        if (!isfinite(li)) flags[2] = 1;
        lambda[rank] = li;
        rank_s[t] = rank;
    }
    const cplx* Lb = Lg;
    if (VLDS && A) {                                        // C is no longer needed: bring L back into its place
        __syncthreads();
        for (int e = t; e < EIG_MAX * EIG_MAX; e += EIG_THREADS) {
            const int i = e & 63, j = e >> 6;
            if (i < r && j < r) C[e] = Lg[e];
        }
        Lb = C;
    }
    __syncthreads();
    if (t < r) {
        // column t of W = L^-H V  (back substitution, in place in V), stored at position rank_s[t]
        const int rank = rank_s[t];
        for (int i = r - 1; i >= 0; --i) {
            cplx x = Vg[i + t * EIG_MAX];
            if (A) {
                for (int k = i + 1; k < r; ++k) x = csub(x, cmul(eig_conj(Lb[k + i * EIG_MAX]), Vg[k + t * EIG_MAX]));
                x = cscale(x, 1.0 / Lb[i + i * EIG_MAX].x);
            }
            Vg[i + t * EIG_MAX] = x;
            Vout[i + (size_t)rank * ld] = x;
        }
    }
}

int fh_launch_herm_eig(int r, int ld, const cplx* S, const cplx* A, void* scratch, double* lambda, cplx* Vout, int* flags,
                       hipStream_t st) {
    cplx* Lg = (cplx*)scratch;
    cplx* Vg = Lg + EIG_MAX * EIG_MAX;
    double* rot = (double*)(Vg + EIG_MAX * EIG_MAX);
    // one-time: ask for the large dynamic LDS allocation; fall back to the global-memory variant if refused
    static int lds_mode = -1;
    const size_t dyn = (size_t)2 * EIG_MAX * EIG_MAX * sizeof(cplx) + 32 * 4 * sizeof(double);
    if (lds_mode < 0) {
        lds_mode = 0;
        if (!getenv("FH_EIG_NO_LDS") &&
            hipFuncSetAttribute((const void*)k_herm_eig<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) == hipSuccess)
            lds_mode = 1;
        (void)hipGetLastError();
    }
    if (lds_mode == 1)
        hipLaunchKernelGGL((k_herm_eig<true>), dim3(1), dim3(EIG_THREADS), dyn, st, r, ld, S, A, Lg, Vg, rot, flags, lambda, Vout, 30);
    else
        hipLaunchKernelGGL((k_herm_eig<false>), dim3(1), dim3(EIG_THREADS), 0, st, r, ld, S, A, Lg, Vg, rot, flags, lambda, Vout, 30);
    return 0;
}

size_t fh_herm_eig_scratch_bytes() { return (size_t)2 * EIG_MAX * EIG_MAX * sizeof(cplx) + 32 * 4 * sizeof(double) + 64; }
