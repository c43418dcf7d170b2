// fh_gmres.hip -- device-resident restarted GMRES(m), batched over quadrature nodes and right-hand-side columns.
//
// Replaces the reference's iterative shifted solve, one column at a time on the CPU:
//   solve_shifted_iterative!   src/sparse/feast_sparse.jl:164-203   (Krylov.jl gmres: restart = true, memory = m,
//   solve_dense_shifted!       src/dense/feast_dense.jl:26-67        zero initial guess, stop ||r|| <= atol + rtol ||r0||)
//
// All columns of all local nodes advance in lock-step.  The Arnoldi basis lives in HBM as panels V[node][0..m], the
// per-(node, column) Hessenberg matrices, Givens rotations and right-hand sides g in device arrays that only these
// kernels touch: inside a restart cycle nothing returns to the host.  Orthogonalisation is classical Gram-Schmidt
// applied twice (CGS2): every pass is ONE multi-dot kernel (W against up to 8 basis panels per block, so W is read
// once per 8 panels) and ONE update kernel, instead of the 2(k+1) kernels of modified Gram-Schmidt; its loss of
// orthogonality is O(eps) like MGS with re-orthogonalisation.  The lane that owns a column owns its scalars.
#include "fh_common.hpp"
#include "fh_kernels.hpp"

#define GM_BLOCK 256
#define GM_CHUNK 8          // basis panels per multi-dot block

// ---- multi-dot: partial[node][chunk][blk][j][c] = sum_rows conj(V_{chunk*8+j}[row, c]) * W[row, c] --------------------
template <int LD>
__global__ __launch_bounds__(GM_BLOCK) void k_gm_dots(fh_gmres_args a, int k) {
    const int node = blockIdx.y, chunk = blockIdx.z;
    if (a.node_active[node] == 0) return;
    const int c = threadIdx.x % LD;
    const int i0 = chunk * GM_CHUNK, ni = min(GM_CHUNK, k + 1 - i0);
    const size_t total = (size_t)a.N * LD;
    const cplx* W = a.W + (size_t)node * a.panel;
    const cplx* V = a.V + (size_t)node * a.v_node_stride + (size_t)i0 * a.panel;
    cplx acc[GM_CHUNK];
#pragma unroll
    for (int j = 0; j < GM_CHUNK; ++j) acc[j] = cmake(0, 0);
    if (a.active[node * LD + c]) {
        for (size_t e = (size_t)blockIdx.x * GM_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * GM_BLOCK) {
            const cplx w = W[e];
#pragma unroll
            for (int j = 0; j < GM_CHUNK; ++j)
                if (j < ni) acc[j] = cadd(acc[j], cmulc(V[(size_t)j * a.panel + e], w));
        }
    }
    __shared__ cplx red[GM_BLOCK];
    const int nchunk = gridDim.z;
    cplx* out = a.partial + ((((size_t)node * nchunk + chunk) * gridDim.x + blockIdx.x) * GM_CHUNK) * LD;
    for (int j = 0; j < ni; ++j) {
        red[threadIdx.x] = acc[j];
        __syncthreads();
        if (threadIdx.x < LD) {
            cplx s = red[threadIdx.x];
            for (int q = 1; q < GM_BLOCK / LD; ++q) s = cadd(s, red[threadIdx.x + q * LD]);
            out[(size_t)j * LD + threadIdx.x] = s;
        }
        __syncthreads();
    }
}

// ---- reduce the multi-dot partials: hcur[node][c][i] (pass 0: =, pass 1: the correction, also added to H) -------------
template <int LD>
__global__ __launch_bounds__(GM_BLOCK) void k_gm_fin_h(fh_gmres_args a, int k, int nblk, int nchunk, int pass) {
    const int node = blockIdx.x;
    if (a.node_active[node] == 0) return;
    __shared__ cplx red[GM_BLOCK];
    const int c = threadIdx.x % LD, grp = threadIdx.x / LD;
    constexpr int G = GM_BLOCK / LD;
    for (int i = 0; i <= k; ++i) {
        const int chunk = i / GM_CHUNK, j = i % GM_CHUNK;
        const cplx* p = a.partial + (((size_t)node * nchunk + chunk) * nblk * GM_CHUNK + j) * LD + c;
        cplx s = cmake(0, 0);
        for (int b = grp; b < nblk; b += G) s = cadd(s, p[(size_t)b * GM_CHUNK * LD]);
        red[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < LD) {
            cplx tot = red[threadIdx.x];
            for (int q = 1; q < G; ++q) tot = cadd(tot, red[threadIdx.x + q * LD]);
            const size_t col = (size_t)node * LD + threadIdx.x;
            if (!a.active[col]) tot = cmake(0, 0);
            a.hcur[col * (a.mr + 1) + i] = tot;
            cplx* Hc = a.H + col * (size_t)(a.mr + 1) * a.mr + (size_t)k * (a.mr + 1);
            Hc[i] = pass == 0 ? tot : cadd(Hc[i], tot);
        }
        __syncthreads();
    }
}

// ---- W -= sum_i hcur_i V_i ; pass 1 also leaves ||W||^2 partials -----------------------------------------------------------
template <int LD>
__global__ __launch_bounds__(GM_BLOCK) void k_gm_update(fh_gmres_args a, int k, int want_norm) {
    const int node = blockIdx.y;
    if (a.node_active[node] == 0) return;
    const int c = threadIdx.x % LD;
    const size_t col = (size_t)node * LD + c;
    const size_t total = (size_t)a.N * LD;
    cplx* W = a.W + (size_t)node * a.panel;
    const cplx* V = a.V + (size_t)node * a.v_node_stride;
    const cplx* hc = a.hcur + col * (a.mr + 1);
    double nrm = 0.0;
    if (a.active[col]) {
        for (size_t e = (size_t)blockIdx.x * GM_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * GM_BLOCK) {
            cplx w = W[e];
            for (int i = 0; i <= k; ++i) {
                const cplx hv = hc[i];
                const cplx v = V[(size_t)i * a.panel + e];
                w.x -= hv.x * v.x - hv.y * v.y;
                w.y -= hv.x * v.y + hv.y * v.x;
            }
            W[e] = w;
            nrm += cabs2(w);
        }
    }
    if (want_norm) {
        __shared__ cplx red[GM_BLOCK];
        red[threadIdx.x] = cmake(nrm, 0);
        __syncthreads();
        if (threadIdx.x < LD) {
            cplx s = red[threadIdx.x];
            for (int q = 1; q < GM_BLOCK / LD; ++q) s = cadd(s, red[threadIdx.x + q * LD]);
            a.npartial[((size_t)node * gridDim.x + blockIdx.x) * LD + threadIdx.x] = s;
        }
    }
}

// ---- cycle start: beta = ||r||, targets (first cycle), g = beta e_1, inv = 1/beta --------------------------------------------
// npartial holds the ||r||^2 partials of the residual product (nblk rows per node).
template <int LD>
__global__ __launch_bounds__(GM_BLOCK) void k_gm_start(fh_gmres_args a, int nblk, int first, double rtol, double atol, int m) {
    const int node = blockIdx.x;
    __shared__ cplx red[GM_BLOCK];
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    const int c = threadIdx.x % LD, grp = threadIdx.x / LD;
    constexpr int G = GM_BLOCK / LD;
    cplx s = cmake(0, 0);
    for (int b = grp; b < nblk; b += G) s = cadd(s, a.npartial[((size_t)node * nblk + b) * LD + c]);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < LD) {
        cplx tot = red[threadIdx.x];
        for (int q = 1; q < G; ++q) tot = cadd(tot, red[threadIdx.x + q * LD]);
        const size_t col = (size_t)node * LD + threadIdx.x;
        const double beta = sqrt(tot.x);
        if (first) {
            a.r0norm[col] = beta;
            a.target[col] = atol + rtol * beta;
            a.iters[col] = 0;
            a.status[col] = 0;
        }
        a.rnorm[col] = beta;
        int act = (threadIdx.x < m) && isfinite(beta) && (beta > a.target[col]);
        if (threadIdx.x < m && !isfinite(beta)) a.status[col] = 8;
        a.active[col] = act;
        a.kdim[col] = 0;
        a.inv[col] = act && beta > 0.0 ? 1.0 / beta : 0.0;
        cplx* g = a.g + col * (a.mr + 1);
        for (int i = 0; i <= a.mr; ++i) g[i] = cmake(0, 0);
        g[0] = cmake(act ? beta : 0.0, 0);
        if (act) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) a.node_active[node] = cnt;
}

// ---- Givens step of column (node, c) after Arnoldi step k (Krylov.jl / Saad: rotations on the new Hessenberg column) ---------
template <int LD>
__global__ __launch_bounds__(GM_BLOCK) void k_gm_givens(fh_gmres_args a, int k, int nblk) {
    const int node = blockIdx.x;
    if (a.node_active[node] == 0) return;
    __shared__ cplx red[GM_BLOCK];
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    const int c = threadIdx.x % LD, grp = threadIdx.x / LD;
    constexpr int G = GM_BLOCK / LD;
    cplx s = cmake(0, 0);
    for (int b = grp; b < nblk; b += G) s = cadd(s, a.npartial[((size_t)node * nblk + b) * LD + c]);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < LD) {
        cplx tot = red[threadIdx.x];
        for (int q = 1; q < G; ++q) tot = cadd(tot, red[threadIdx.x + q * LD]);
        const size_t col = (size_t)node * LD + threadIdx.x;
        if (a.active[col]) {
            const double hk1 = sqrt(tot.x);
            cplx* Hc = a.H + col * (size_t)(a.mr + 1) * a.mr + (size_t)k * (a.mr + 1);
            cplx* cs = a.cs + col * a.mr;
            cplx* sn = a.sn + col * a.mr;
            cplx* g = a.g + col * (a.mr + 1);
            Hc[k + 1] = cmake(hk1, 0);
            for (int i = 0; i < k; ++i) {
                const cplx t = cadd(cmul(cs[i], Hc[i]), cmul(sn[i], Hc[i + 1]));
                Hc[i + 1] = cadd(cmul(cmake(-sn[i].x, sn[i].y), Hc[i]), cmul(cs[i], Hc[i + 1]));
                Hc[i] = t;
            }
            const cplx av = Hc[k], bv = Hc[k + 1];
            const double aa = sqrt(cabs2(av)), den = sqrt(cabs2(av) + cabs2(bv));
            if (den == 0.0) { cs[k] = cmake(1, 0); sn[k] = cmake(0, 0); }
            else if (aa == 0.0) { cs[k] = cmake(0, 0); sn[k] = cmake(1, 0); }
            else {
                cs[k] = cmake(aa / den, 0);
                sn[k] = cscale(cmul(cscale(av, 1.0 / aa), cconj(bv)), 1.0 / den);
            }
            Hc[k] = cadd(cmul(cs[k], av), cmul(sn[k], bv));
            Hc[k + 1] = cmake(0, 0);
            g[k + 1] = cmul(cmake(-sn[k].x, sn[k].y), g[k]);
            g[k] = cmul(cs[k], g[k]);
            a.iters[col] += 1;
            a.kdim[col] = k + 1;
            const double rn = sqrt(cabs2(g[k + 1]));
            a.rnorm[col] = rn;
            int act = 1;
            if (!isfinite(rn) || !isfinite(hk1)) { act = 0; a.status[col] = 8; }
            else if (!(rn > a.target[col]) || hk1 == 0.0) act = 0;            // converged, or lucky breakdown
            a.active[col] = act;
            a.inv[col] = act ? 1.0 / hk1 : 0.0;
            if (act) atomicAdd(&cnt, 1);
        } else {
            a.inv[col] = 0.0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) a.node_active[node] = cnt;
}

// ---- V_dst = src * inv[col]   (v_0 = r / beta ; v_{k+1} = w / h_{k+1,k}; zero for finished columns) ------------------------------
template <int LD>
__global__ __launch_bounds__(GM_BLOCK) void k_gm_scale_store(fh_gmres_args a, const cplx* src, size_t src_node_stride, int dst_index) {
    const int node = blockIdx.y;
    const int c = threadIdx.x % LD;
    const double f = a.inv[(size_t)node * LD + c];
    const size_t total = (size_t)a.N * LD;
    const cplx* S = src + (size_t)node * src_node_stride;
    cplx* D = a.V + (size_t)node * a.v_node_stride + (size_t)dst_index * a.panel;
    for (size_t e = (size_t)blockIdx.x * GM_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * GM_BLOCK) {
        const cplx v = S[e];
        D[e] = cmake(v.x * f, v.y * f);
    }
}

// ---- end of cycle: y from the triangular systems (one lane per column), then X += V y --------------------------------------------
template <int LD>
__global__ __launch_bounds__(GM_BLOCK) void k_gm_solve_y(fh_gmres_args a, int nodes) {
    const int idx = blockIdx.x * GM_BLOCK + threadIdx.x;
    if (idx >= nodes * LD) return;
    const size_t col = (size_t)idx;
    const int kk = a.kdim[col];
    const cplx* H = a.H + col * (size_t)(a.mr + 1) * a.mr;
    const cplx* g = a.g + col * (a.mr + 1);
    cplx* y = a.y + col * a.mr;
    for (int i = 0; i < a.mr; ++i) y[i] = cmake(0, 0);
    for (int i = kk - 1; i >= 0; --i) {
        cplx s = g[i];
        for (int j = i + 1; j < kk; ++j) s = csub(s, cmul(H[(size_t)j * (a.mr + 1) + i], y[j]));
        const cplx d = H[(size_t)i * (a.mr + 1) + i];
        y[i] = cabs2(d) > 0 ? cdiv(s, d) : cmake(0, 0);
    }
}

template <int LD>
__global__ __launch_bounds__(GM_BLOCK) void k_gm_xupdate(fh_gmres_args a, cplx* X, size_t x_node_stride, int kmax) {
    const int node = blockIdx.y;
    const int c = threadIdx.x % LD;
    const size_t col = (size_t)node * LD + c;
    const int kk = min(kmax, a.kdim[col]);
    if (kk == 0) return;
    const cplx* y = a.y + col * a.mr;
    const cplx* V = a.V + (size_t)node * a.v_node_stride;
    cplx* Xn = X + (size_t)node * x_node_stride;
    const size_t total = (size_t)a.N * LD;
    for (size_t e = (size_t)blockIdx.x * GM_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * GM_BLOCK) {
        cplx x = Xn[e];
        for (int i = 0; i < kk; ++i) cfma(x, y[i], V[(size_t)i * a.panel + e]);
        Xn[e] = x;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
#define GM_DISPATCH(ld, KERNEL, grid, st, ...)                                                        \
    do {                                                                                               \
        if ((ld) == 16) hipLaunchKernelGGL((KERNEL<16>), grid, dim3(GM_BLOCK), 0, st, __VA_ARGS__);     \
        else if ((ld) == 32) hipLaunchKernelGGL((KERNEL<32>), grid, dim3(GM_BLOCK), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<64>), grid, dim3(GM_BLOCK), 0, st, __VA_ARGS__);                \
    } while (0)

int fh_gm_nchunk(int k) { return (k + 1 + GM_CHUNK - 1) / GM_CHUNK; }
size_t fh_gm_partial_elems(int mr, int nblk, int nodes, int ld) { return (size_t)nodes * fh_gm_nchunk(mr) * nblk * GM_CHUNK * ld; }

void fh_launch_gm_orthogonalize(const fh_gmres_args& a, int ld, int k, int nblk, int nodes, hipStream_t st) {
    const int nchunk = fh_gm_nchunk(k);
    for (int pass = 0; pass < 2; ++pass) {
        GM_DISPATCH(ld, k_gm_dots, dim3(nblk, nodes, nchunk), st, a, k);
        GM_DISPATCH(ld, k_gm_fin_h, dim3(nodes), st, a, k, nblk, nchunk, pass);
        GM_DISPATCH(ld, k_gm_update, dim3(nblk, nodes), st, a, k, pass);
    }
}
void fh_launch_gm_start(const fh_gmres_args& a, int ld, int nblk_norm, int nodes, int first, double rtol, double atol, int m, hipStream_t st) {
    GM_DISPATCH(ld, k_gm_start, dim3(nodes), st, a, nblk_norm, first, rtol, atol, m);
}
void fh_launch_gm_givens(const fh_gmres_args& a, int ld, int k, int nblk, int nodes, hipStream_t st) {
    GM_DISPATCH(ld, k_gm_givens, dim3(nodes), st, a, k, nblk);
}
void fh_launch_gm_scale_store(const fh_gmres_args& a, int ld, const cplx* src, size_t src_node_stride, int dst_index, int nblk, int nodes,
                              hipStream_t st) {
    GM_DISPATCH(ld, k_gm_scale_store, dim3(nblk, nodes), st, a, src, src_node_stride, dst_index);
}
void fh_launch_gm_finish_cycle(const fh_gmres_args& a, int ld, cplx* X, size_t x_node_stride, int kmax, int nblk, int nodes, hipStream_t st) {
    GM_DISPATCH(ld, k_gm_solve_y, dim3((nodes * ld + GM_BLOCK - 1) / GM_BLOCK), st, a, nodes);
    GM_DISPATCH(ld, k_gm_xupdate, dim3(nblk, nodes), st, a, X, x_node_stride, kmax);
}
