// fh_api.hip -- the C ABI of libfeasthip.so (include/feasthip.h).  Host-side orchestration
// of the gfx950 kernels; no arithmetic on the host except M0 x M0 bookkeeping.
#include "fh_common.hpp"
#include "fh_kernels.hpp"
#include "fh_dense.hpp"
#include "fh_banded.hpp"
#include "fh_eig.hpp"
#include "fh_comm.hpp"
#include "../../include/feasthip.h"

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstring>

// ---------------------------------------------------------------------------------------
// workspace + profiling helpers
// ---------------------------------------------------------------------------------------
int fh_get_buf(feasthip_ctx* h, const char* name, size_t bytes, void** out) {
    auto it = h->bufs.find(name);
    if (it != h->bufs.end() && it->second.second >= bytes) {
        *out = it->second.first;
        return 0;
    }
    if (it != h->bufs.end()) {
        hipStreamSynchronize(h->stream);
        hipFree(it->second.first);
        h->bufs.erase(it);
    }
    void* p = nullptr;
    size_t alloc = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&p, alloc);
    if (e != hipSuccess) {
        h->last_error = std::string("hipMalloc(") + name + ", " + std::to_string(alloc) + "): " + hipGetErrorString(e);
        return FEASTHIP_ERROR_MEMORY;
    }
    h->bufs[name] = {p, alloc};
    *out = p;
    return 0;
}

// A call is about to return an error while its kernels may still be queued or running on h->stream (progress
// deadline, faulted queue).  Give the stream a bounded chance to drain; if it does not, the handle is POISONED: the
// workspaces those kernels use must not be reused or freed, so every later call fails fast until destroy (which then
// skips the stream synchronisation and leaks the buffers to the process -- the host must exit non-zero or continue in
// a fresh process; see include/feasthip.h).
static void fh_poison_unless_drained(feasthip_ctx* h, double grace_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) return;                        // drained: the handle stays usable
        if (q != hipErrorNotReady) break;                   // the queue itself reports a fault
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > grace_s) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    h->poisoned = 1;
    h->last_error += " [handle poisoned: device work may still be in flight; destroy the handle and exit the process]";
}

void fh_free_bufs(feasthip_ctx* h) {
    for (auto& kv : h->bufs) hipFree(kv.second.first);
    h->bufs.clear();
}

// Sampled event timing (1 launch in 13: a period coprime to the iteration caps, so the samples do not alias with
// the position inside a solve, where kernel durations shrink as nodes converge): every FH_PROF_PERIOD-th launch of a class is bracketed by two events
// on the launch stream; the class average is (sum of sampled durations)/(samples).
#define FH_PROF_PERIOD 13
static thread_local int fh_prof_open = 0;
static int fh_prof_period() {
    static const int period = getenv("FH_PROF_PERIOD") ? std::max(1, atoi(getenv("FH_PROF_PERIOD"))) : FH_PROF_PERIOD;
    return period;
}
static inline double fh_now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
void fh_prof_begin(feasthip_ctx* h, const char* cls) {
    fh_prof_open = 0;
    if (!h->profiling) return;
    fh_prof_class& pc = h->prof[cls];
    pc.launches += 1;
    const int period = h->prof_period > 0 ? h->prof_period : fh_prof_period();
    static const bool nopool = getenv("FH_PROF_NOPOOL") != nullptr;
    const long eff = (long)period * h->prof_mult;
    if (eff > 1 && (pc.launches % eff) != 1) return;
    if (h->pending_events.size() > 60000) return;
    const double t_in = fh_now_s();
    fh_event_pair ep;
    ep.cls = cls;
    // events are recycled through a pool: creating and destroying a pair per sample cost more than recording it
    if (!nopool && h->event_pool.size() >= 2) {
        ep.a = h->event_pool.back(); h->event_pool.pop_back();
        ep.b = h->event_pool.back(); h->event_pool.pop_back();
    } else {
        if (hipEventCreate(&ep.a) != hipSuccess) return;
        if (hipEventCreate(&ep.b) != hipSuccess) { hipEventDestroy(ep.a); return; }
    }
    hipEventRecord(ep.a, h->stream);
    h->pending_events.push_back(ep);
    fh_prof_open = 1;
    h->prof_host_s += fh_now_s() - t_in;
}
void fh_prof_end(feasthip_ctx* h) {
    if (!fh_prof_open) return;
    const double t_in = fh_now_s();
    hipEventRecord(h->pending_events.back().b, h->stream);
    fh_prof_open = 0;
    h->prof_host_s += fh_now_s() - t_in;
}
void fh_prof_collect(feasthip_ctx* h) {
    if (h->pending_events.empty()) return;
    hipStreamSynchronize(h->stream);
    const double t_in = fh_now_s();
    for (auto& ep : h->pending_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            fh_prof_class& pc = h->prof[ep.cls + std::string("#sampled")];
            pc.total_ms += ms;
            pc.launches += 1;
        }
        if (h->event_pool.size() < 8192) { h->event_pool.push_back(ep.a); h->event_pool.push_back(ep.b); }
        else { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
    }
    h->pending_events.clear();
    // Event calls are cheap on most hosts (0.7 % of a bench step at 1 launch in 13) but were seen to cost ~90 us each
    // on a loaded box: when sampling has taken more than 1 % of the wall time since it was switched on, sample 7x
    // less often (91 stays coprime to the iteration caps), up to 1 launch in 637.
    h->prof_host_s += fh_now_s() - t_in;
    const double wall = fh_now_s() - h->prof_t0;
    if ((h->prof_period > 0 ? h->prof_period : fh_prof_period()) > 1 && wall > 0.05 && h->prof_host_s > 0.01 * wall && h->prof_mult < 49) {   // period 1 = exact timing requested
        h->prof_mult *= 7;
        h->prof_host_s = 0.0;
        h->prof_t0 = fh_now_s();
    }
}

// ---------------------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------------------
extern "C" int feasthip_version(int* major, int* minor) {
    if (major) *major = FEASTHIP_VERSION_MAJOR;
    if (minor) *minor = FEASTHIP_VERSION_MINOR;
    return 0;
}

extern "C" int feasthip_create(feasthip_handle* out, int device_id) {
    if (!out) return FEASTHIP_ERROR_INTERNAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FEASTHIP_ERROR_INTERNAL;
    if (device_id < 0 || device_id >= ndev) return FEASTHIP_ERROR_INTERNAL;
    feasthip_ctx* h = new (std::nothrow) feasthip_ctx();
    if (!h) return FEASTHIP_ERROR_MEMORY;
    h->device = device_id;
    if (getenv("FH_LU_KB")) h->lu_outer_block = std::max(32, (atoi(getenv("FH_LU_KB")) / 32) * 32);
    h->lu_solve_legacy = getenv("FH_LU_SOLVE_32") != nullptr;
    h->lu_gemm_staged = getenv("FH_LU_GEMM_STAGED") != nullptr;
    if (getenv("FH_LU_LOOKAHEAD")) h->lu_lookahead = atoi(getenv("FH_LU_LOOKAHEAD"));
    h->sum_mode = getenv("FH_NO_SUM_MODE") ? 0 : 1;
    h->lu_panel_legacy = getenv("FH_LU_PANEL_LEGACY") ? atoi(getenv("FH_LU_PANEL_LEGACY")) : 0;
    if (hipSetDevice(device_id) != hipSuccess) { delete h; return FEASTHIP_ERROR_INTERNAL; }
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) { delete h; return FEASTHIP_ERROR_INTERNAL; }
    h->stream = h->own_stream;
    if (hipHostMalloc((void**)&h->pin, (size_t)1 << 20, hipHostMallocDefault) == hipSuccess) h->pin_cap = (size_t)1 << 20;
    else { h->pin = nullptr; h->pin_cap = 0; hipGetLastError(); }
    if (hipMalloc((void**)&h->d_counters, 8 * sizeof(unsigned long long)) != hipSuccess) { hipStreamDestroy(h->own_stream); delete h; return FEASTHIP_ERROR_MEMORY; }
    hipMemset(h->d_counters, 0, 8 * sizeof(unsigned long long));
    {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess) { hipFree(h->d_counters); hipStreamDestroy(h->own_stream); delete h; return FEASTHIP_ERROR_MEMORY; }
        h->h_progress = (volatile unsigned long long*)hp;
        *h->h_progress = 0ull;
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { hipHostFree(hp); hipFree(h->d_counters); hipStreamDestroy(h->own_stream); delete h; return FEASTHIP_ERROR_INTERNAL; }
        h->d_progress = (unsigned long long*)dp;
    }
    *out = h;
    return 0;
}

static void fh_free_problem(feasthip_ctx* h) {
    if (h->csr.rowptr) hipFree(h->csr.rowptr);
    if (h->csr.col) hipFree(h->csr.col);
    if (h->csr.aval) hipFree(h->csr.aval);
    if (h->csr.bval) hipFree(h->csr.bval);
    if (h->csr.perm) hipFree(h->csr.perm);
    if (h->csr.rp8) hipFree(h->csr.rp8);
    if (h->csr.col8) hipFree(h->csr.col8);
    if (h->csr.a8) hipFree(h->csr.a8);
    if (h->csr.b8) hipFree(h->csr.b8);
    if (h->csr.blk_start) hipFree(h->csr.blk_start);
    if (h->csr.ext_ptr) hipFree(h->csr.ext_ptr);
    if (h->csr.ext_idx) hipFree(h->csr.ext_idx);
    if (h->csr.lcol) hipFree(h->csr.lcol);
    h->csr = fh_csr();
    if (h->dense.A) hipFree(h->dense.A);
    if (h->dense.B) hipFree(h->dense.B);
    h->dense = fh_dense();
    for (void* p : h->lu_factors) if (p) hipFree(p);
    for (int* p : h->lu_pivots) if (p) hipFree(p);
    h->lu_factors.clear(); h->lu_pivots.clear(); h->lu_valid.clear(); h->lu_z.clear();
    fh_banded_free(h);
    h->kind = 0;
    h->rs_P = h->rs_basis = h->rs_X = h->rs_R = nullptr;
    h->rs_m = h->rs_ld = h->rs_rank = h->rs_X_m = h->rs_X_ld = 0;
    h->rs_T.clear(); h->rs_R_lambda.clear();
}

extern "C" int feasthip_destroy(feasthip_handle h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    if (h->poisoned) {
        // kernels of a failed call may still be running on the workspaces: neither wait for them (a wedged kernel
        // never ends) nor free what they touch.  The memory goes back to the driver when the process exits.
        fh_comm_mark_failed(h);
        delete h;
        return 0;
    }
    hipStreamSynchronize(h->stream);
    fh_comm_destroy(h);
    fh_prof_collect(h);
    fh_free_problem(h);
    fh_free_bufs(h);
    if (h->d_counters) hipFree(h->d_counters);
    for (auto& ep : h->pending_events) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
    for (hipEvent_t e : h->event_pool) hipEventDestroy(e);
    if (h->h_progress) hipHostFree((void*)h->h_progress);
    if (h->pin) hipHostFree(h->pin);
    if (h->lu_ev_next) hipEventDestroy(h->lu_ev_next);
    if (h->lu_ev_rest) hipEventDestroy(h->lu_ev_rest);
    if (h->side_stream) hipStreamDestroy(h->side_stream);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return 0;
}

extern "C" const char* feasthip_last_error(feasthip_handle h) { return h ? h->last_error.c_str() : "null handle"; }

extern "C" int feasthip_set_stream(feasthip_handle h, void* hip_stream) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (h->poisoned) { h->last_error = "handle poisoned by an earlier device failure: destroy it"; return FEASTHIP_ERROR_INTERNAL; }
    hipStreamSynchronize(h->stream);
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return 0;
}

extern "C" int feasthip_synchronize(feasthip_handle h) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (h->poisoned) { h->last_error = "handle poisoned by an earlier device failure: destroy it"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    FH_CHECK(hipStreamSynchronize(h->stream));
    fh_prof_collect(h);
    FH_CHECK(hipGetLastError());       // launch-configuration errors do not surface through the stream sync
    return 0;
}

// ---------------------------------------------------------------------------------------
// problem definition
// ---------------------------------------------------------------------------------------
static inline cplx fh_ing_zero(cplx) { return cmake(0, 0); }
static inline cplx fh_ing_add(cplx a, cplx b) { return cadd(a, b); }
#define FH_INGEST_STORAGE_CSR FEASTHIP_STORAGE_CSR
#include "fh_ingest.hpp"       // host side of the ingest (pure C++; also compiled under ASan/UBSan by tests/host_ingest_harness.cpp)
#include "fh_policy.hpp"       // host policy of the inexact mode (pure C++), exported as feasthip_policy_*

template <typename VT>
static int set_csr_typed(feasthip_ctx* h, int64_t N, int index_base, int storage, int64_t nnzA, const int64_t* ptrA,
                         const int64_t* idxA, const VT* valA, int64_t nnzB, const int64_t* ptrB, const int64_t* idxB,
                         const VT* valB) {
    // Row-block renumbering (fh_ingest.hpp: recursive bisection into 128-row blocks).  ON by default for wide patterns since
    // round 3: the row-per-wave SpMM is bound by the traffic that misses L2 (1-KB rows: the two far stencil planes of the
    // caller's order do not fit 4 MiB), and compact blocks keep a slice's gathers local -- cfg 3: 37 -> 30 us per node and
    // launch, 163 -> 159 ms per solve.  (k_spmm, with its 256-B tiles, never cared: round 2.)  FH_REORDER=0 keeps the caller's
    // order, FH_REORDER=2 renumbers whenever there are at least two blocks (test rigs push small problems through it).  The
    // LDS-window kernel the renumbering was built for stays opt-in (FH_LDS_SPMM=1): 57 vs 33 us per node on cfg 3.
    static const int reorder_mode = getenv("FH_REORDER") ? atoi(getenv("FH_REORDER")) : 1;
    fh_prepared<VT> P;
    std::string err;
    const int prc = fh_prepare_csr<VT>(N, index_base, storage, nnzA, ptrA, idxA, valA, nnzB, ptrB, idxB, valB, reorder_mode,
                                       FH_SPMM_R, FH_SPMM_EXT, sizeof(VT) == sizeof(double), P, err);
    if (prc) { h->last_error = "feasthip_set_csr: " + err; return prc == 3 ? FEASTHIP_ERROR_MEMORY : FEASTHIP_ERROR_N; }
    const bool hasB = ptrB != nullptr;
    fh_free_problem(h);
    h->csr_kl = P.kl; h->csr_ku = P.ku;
    h->host_rowptr = P.rowptr; h->host_col = P.col;
    const std::vector<int>& rowptr = P.rowptr;
    const std::vector<int>& col = P.col;
    fh_csr& d = h->csr;
    d.N = N; d.nnz = (int64_t)col.size(); d.is_complex = sizeof(VT) == sizeof(cplx); d.b_identity = hasB ? 0 : 1;
    FH_CHECK(hipMalloc((void**)&d.rowptr, (N + 1) * sizeof(int)));
    FH_CHECK(hipMalloc((void**)&d.col, std::max<size_t>(1, col.size()) * sizeof(int)));
    FH_CHECK(hipMalloc(&d.aval, std::max<size_t>(1, col.size()) * sizeof(VT)));
    FH_CHECK(hipMemcpy(d.rowptr, rowptr.data(), (N + 1) * sizeof(int), hipMemcpyHostToDevice));
    FH_CHECK(hipMemcpy(d.col, col.data(), col.size() * sizeof(int), hipMemcpyHostToDevice));
    FH_CHECK(hipMemcpy(d.aval, P.av.data(), col.size() * sizeof(VT), hipMemcpyHostToDevice));
    if (hasB) {
        FH_CHECK(hipMalloc(&d.bval, std::max<size_t>(1, col.size()) * sizeof(VT)));
        FH_CHECK(hipMemcpy(d.bval, P.bv.data(), col.size() * sizeof(VT), hipMemcpyHostToDevice));
    }
    if (!P.rp8.empty()) {
        FH_CHECK(hipMalloc((void**)&d.rp8, (N + 1) * sizeof(int)));
        FH_CHECK(hipMalloc((void**)&d.col8, P.col8.size() * sizeof(int)));
        FH_CHECK(hipMalloc((void**)&d.a8, P.a8.size() * sizeof(double)));
        FH_CHECK(hipMemcpy(d.rp8, P.rp8.data(), (N + 1) * sizeof(int), hipMemcpyHostToDevice));
        FH_CHECK(hipMemcpy(d.col8, P.col8.data(), P.col8.size() * sizeof(int), hipMemcpyHostToDevice));
        FH_CHECK(hipMemcpy(d.a8, P.a8.data(), P.a8.size() * sizeof(double), hipMemcpyHostToDevice));
        if (hasB) {
            FH_CHECK(hipMalloc((void**)&d.b8, P.b8.size() * sizeof(double)));
            FH_CHECK(hipMemcpy(d.b8, P.b8.data(), P.b8.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    if (!P.perm.empty()) {
        const int nb = (int)P.blk_start.size() - 1;
        FH_CHECK(hipMalloc((void**)&d.perm, N * sizeof(int)));
        FH_CHECK(hipMemcpy(d.perm, P.perm.data(), N * sizeof(int), hipMemcpyHostToDevice));
        d.nblk = nb;
        FH_CHECK(hipMalloc((void**)&d.blk_start, (nb + 1) * sizeof(int)));
        FH_CHECK(hipMalloc((void**)&d.ext_ptr, (nb + 1) * sizeof(int)));
        FH_CHECK(hipMalloc((void**)&d.ext_idx, std::max<size_t>(1, P.ext_idx.size()) * sizeof(int)));
        FH_CHECK(hipMalloc((void**)&d.lcol, std::max<size_t>(1, P.lcol.size()) * sizeof(unsigned short)));
        FH_CHECK(hipMemcpy(d.blk_start, P.blk_start.data(), (nb + 1) * sizeof(int), hipMemcpyHostToDevice));
        FH_CHECK(hipMemcpy(d.ext_ptr, P.ext_ptr.data(), (nb + 1) * sizeof(int), hipMemcpyHostToDevice));
        FH_CHECK(hipMemcpy(d.ext_idx, P.ext_idx.data(), P.ext_idx.size() * sizeof(int), hipMemcpyHostToDevice));
        FH_CHECK(hipMemcpy(d.lcol, P.lcol.data(), P.lcol.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        if (getenv("FH_DEBUG_TIMING"))
            fprintf(stderr, "[feasthip] renumbered into %d row blocks, %.1f outside rows per block on average\n", nb, nb ? (double)P.ext_idx.size() / nb : 0.0);
    }
    h->kind = 2;
    return 0;
}

extern "C" int feasthip_set_csr(feasthip_handle h, int64_t N, int is_complex, int index_base, int storage,
                                int64_t nnzA, const int64_t* ptrA, const int64_t* idxA, const void* valA,
                                int64_t nnzB, const int64_t* ptrB, const int64_t* idxB, const void* valB) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (N <= 0 || N > INT32_MAX / FH_MAX_LD) { h->last_error = "feasthip_set_csr: N out of range"; return FEASTHIP_ERROR_N; }
    if (!ptrA || (nnzA > 0 && (!idxA || !valA))) { h->last_error = "feasthip_set_csr: null A"; return FEASTHIP_ERROR_N; }
    if (index_base != 0 && index_base != 1) { h->last_error = "feasthip_set_csr: index_base must be 0 or 1"; return FEASTHIP_ERROR_FPM; }
    if (storage != FEASTHIP_STORAGE_CSR && storage != FEASTHIP_STORAGE_CSC) { h->last_error = "feasthip_set_csr: bad storage"; return FEASTHIP_ERROR_FPM; }
    FH_CHECK(hipSetDevice(h->device));
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (is_complex)
        return set_csr_typed<cplx>(h, N, index_base, storage, nnzA, ptrA, idxA, (const cplx*)valA, nnzB, ptrB, idxB, (const cplx*)valB);
    return set_csr_typed<double>(h, N, index_base, storage, nnzA, ptrA, idxA, (const double*)valA, nnzB, ptrB, idxB, (const double*)valB);
}

extern "C" int feasthip_release_factors(feasthip_handle h) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (h->poisoned) { h->last_error = "handle poisoned by an earlier device failure: destroy it"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (h->side_stream) FH_CHECK(hipStreamSynchronize(h->side_stream));
    for (void* p : h->lu_factors) if (p) hipFree(p);
    for (int* p : h->lu_pivots) if (p) hipFree(p);
    h->lu_factors.clear(); h->lu_pivots.clear(); h->lu_valid.clear(); h->lu_z.clear();
    for (void* p : h->band_factors) if (p) hipFree(p);
    for (int* p : h->band_pivots) if (p) hipFree(p);
    h->band_factors.clear(); h->band_pivots.clear(); h->band_valid.clear(); h->band_z.clear();
    // the multifrontal solver's transient buffers (work arena, substitution panels: together more than the factors themselves)
    for (const char* name : {"mf_work", "mf_y", "mf_z", "mf_ptrs", "mf_info"}) {
        auto it = h->bufs.find(name);
        if (it != h->bufs.end()) { hipFree(it->second.first); h->bufs.erase(it); }
    }
    return 0;
}

extern "C" int feasthip_band_plan(feasthip_handle h, int* kl, int* ku, int64_t* bytes_per_node, int* blocked) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (h->poisoned) { h->last_error = "handle poisoned by an earlier device failure: destroy it"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    return fh_banded_plan(h, kl, ku, bytes_per_node, blocked);
}
extern "C" int feasthip_direct_plan_flops(feasthip_handle h, double* flops_per_node) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (h->poisoned) { h->last_error = "handle poisoned by an earlier device failure: destroy it"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    return fh_banded_plan_flops(h, flops_per_node);
}

extern "C" int feasthip_set_dense(feasthip_handle h, int64_t N, int is_complex, const void* A, int64_t lda,
                                  const void* B, int64_t ldb) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (N <= 0 || N > 65536) { h->last_error = "feasthip_set_dense: N out of range"; return FEASTHIP_ERROR_N; }
    if (!A || lda < N || (B && ldb < N)) { h->last_error = "feasthip_set_dense: bad A/lda/ldb"; return FEASTHIP_ERROR_N; }
    FH_CHECK(hipSetDevice(h->device));
    FH_CHECK(hipStreamSynchronize(h->stream));
    fh_free_problem(h);
    size_t es = is_complex ? sizeof(cplx) : sizeof(double);
    fh_dense& d = h->dense;
    d.N = N; d.is_complex = is_complex; d.b_identity = B ? 0 : 1;
    FH_CHECK(hipMalloc(&d.A, (size_t)N * N * es));
    FH_CHECK(hipMemcpy2D(d.A, (size_t)N * es, A, (size_t)lda * es, (size_t)N * es, (size_t)N, hipMemcpyHostToDevice));
    if (B) {
        FH_CHECK(hipMalloc(&d.B, (size_t)N * N * es));
        FH_CHECK(hipMemcpy2D(d.B, (size_t)N * es, B, (size_t)ldb * es, (size_t)N * es, (size_t)N, hipMemcpyHostToDevice));
    }
    h->kind = 1;
    return 0;
}

extern "C" int feasthip_set_contour(feasthip_handle h, int ne, const double* zne, const double* wne, double weight_scale) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (ne <= 0 || !zne || !wne) { h->last_error = "feasthip_set_contour: ne <= 0 or null arrays"; return FEASTHIP_ERROR_FPM; }
    h->zne.resize(ne); h->wne.resize(ne);
    for (int e = 0; e < ne; ++e) {
        h->zne[e] = cmake(zne[2 * e], zne[2 * e + 1]);
        h->wne[e] = cmake(wne[2 * e], wne[2 * e + 1]);
    }
    h->weight_scale = weight_scale;
    h->node_first = 0;
    h->node_count = ne;
    h->node_ids.resize(ne);
    for (int e = 0; e < ne; ++e) h->node_ids[e] = e;
    // cached factors stay: slot e is reused only when its shift equals the new z_e exactly (fh_dense_lu_solve_nodes,
    // fh_banded_solve_nodes), so a repeated solve on the same contour keeps its factorisations and any other contour
    // refactors slot by slot
    return 0;
}

extern "C" int feasthip_set_real_projection(feasthip_handle h, int real_part) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    h->real_projection = real_part ? 1 : 0;
    return 0;
}

extern "C" int feasthip_set_node_range(feasthip_handle h, int first, int count) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (first < 0 || count < 0 || first + count > (int)h->zne.size()) {
        h->last_error = "feasthip_set_node_range: range outside the contour";
        return FEASTHIP_ERROR_FPM;
    }
    h->node_first = first;
    h->node_count = count;
    h->node_ids.resize(count);
    for (int e = 0; e < count; ++e) h->node_ids[e] = first + e;
    return 0;
}

extern "C" int feasthip_set_node_list(feasthip_handle h, int count, const int* indices) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (count < 0 || (count > 0 && !indices)) { h->last_error = "feasthip_set_node_list: bad arguments"; return FEASTHIP_ERROR_FPM; }
    for (int e = 0; e < count; ++e)
        if (indices[e] < 0 || indices[e] >= (int)h->zne.size()) {
            h->last_error = "feasthip_set_node_list: index outside the contour";
            return FEASTHIP_ERROR_FPM;
        }
    h->node_ids.assign(indices, indices + count);
    h->node_first = count > 0 ? indices[0] : 0;
    h->node_count = count;
    return 0;
}

extern "C" int feasthip_set_column_mask(feasthip_handle h, int64_t m, const int* mask) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (m < 0 || m > (1 << 20)) { h->last_error = "set_column_mask: m out of range"; return FEASTHIP_ERROR_M0; }
    h->col_mask.clear();
    if (mask && m > 0) h->col_mask.assign(mask, mask + m);
    return 0;
}

extern "C" int feasthip_set_solver(feasthip_handle h, int kind, double rtol, double atol, int maxit, int restart,
                                   int factor_precision, int cache_factors) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (kind < 0 || kind > 4 || rtol < 0 || atol < 0 || maxit <= 0 || restart < 0 ||
        (factor_precision != 64 && factor_precision != 32)) {
        h->last_error = "feasthip_set_solver: invalid option";
        return FEASTHIP_ERROR_FPM;
    }
    h->solver = kind; h->rtol = rtol; h->atol = atol; h->maxit = maxit; h->restart = restart;
    h->factor_precision = factor_precision; h->cache_factors = cache_factors;
    return 0;
}

// ---------------------------------------------------------------------------------------
// operator application on panels (sparse or dense):  Y = (cb*B + ca*A) X  per column
// ---------------------------------------------------------------------------------------

struct fh_op_call {
    const void* X; size_t x_stride;
    void* Y; size_t y_stride;
    const cplx* coefA; const cplx* coefB;   // device [nodes][ld]
    const void* Bvec; size_t b_stride;
    const void* U; size_t u_stride;
    int dot_mode; cplx* partial1; cplx* partial2;
    const int* node_active;
    int nodes;
    int m = FH_MAX_LD;     // active columns (measurement only)
    int uniform_coef = 0;  // coefA/coefB identical across columns
    int prec = 64;         // panel precision of X/Y/Bvec/U
    const cplx* colscale = nullptr;   // row kernel only (fh_spmm_args::colscale): X is a shared panel times per-node column factors
};

// full-width panels over a real matrix go through the row-per-wave kernel (FH_SPMM_ROW=0: the 4-rows-per-wave gather kernel)
static bool fh_row_kernel_ok(feasthip_ctx* h, int ld) {
    static const bool row_off = getenv("FH_SPMM_ROW") && atoi(getenv("FH_SPMM_ROW")) == 0;
    return h->kind == 2 && ld == 64 && !h->csr.is_complex && h->csr.rp8 && !row_off;
}

// returns number of blocks used in x (needed to size / read partials)
static int fh_apply_operator(feasthip_ctx* h, int ld, const fh_op_call& c) {
    if (h->kind == 2) {
        fh_spmm_args a;
        a.rowptr = h->csr.rowptr; a.col = h->csr.col; a.aval = h->csr.aval; a.bval = h->csr.bval;
        a.N = (int)h->csr.N; a.nodes = c.nodes;
        a.X = c.X; a.x_node_stride = c.x_stride; a.Y = c.Y; a.y_node_stride = c.y_stride;
        a.coefA = c.coefA; a.coefB = c.coefB; a.Bvec = c.Bvec; a.b_node_stride = c.b_stride;
        a.U = c.U; a.u_node_stride = c.u_stride; a.dot_mode = c.dot_mode;
        a.partial1 = c.partial1; a.partial2 = c.partial2; a.node_active = c.node_active;
        a.counters = h->profiling ? h->d_counters : nullptr; a.m = c.m; a.uniform_coef = c.uniform_coef; a.prec = c.prec;
        static const bool no_lds = !(getenv("FH_LDS_SPMM") && atoi(getenv("FH_LDS_SPMM")) != 0);      // opt-in
        // (the LDS-window kernel keeps its active-node list in a 64-entry LDS array: wider node batches -- trapezoid
        //  contours put no bound on fpm[2] -- take the gather kernel, which has no such limit)
        const bool lds_kernel = h->csr.lcol && c.prec == 64 && !no_lds && c.nodes <= 64 && c.dot_mode != 6;   // (fused-COCG dots: gather kernel only)
        a.nblk_rows = h->csr.nblk; a.blk_start = h->csr.blk_start; a.ext_ptr = h->csr.ext_ptr; a.ext_idx = h->csr.ext_idx;
        a.lcol = lds_kernel ? h->csr.lcol : nullptr;
        // full-width panels over a real matrix: the row-per-wave kernel (FH_SPMM_ROW=0: the 4-rows-per-wave gather kernel)
        a.use_row_kernel = (fh_row_kernel_ok(h, ld) && !lds_kernel) ? 1 : 0;
        a.colscale = c.colscale;
        if (c.colscale && (lds_kernel || c.prec != 64 || h->csr.is_complex)) { h->last_error = "internal: column-scaled operand needs a gather kernel over a real matrix on complex128 panels"; return -1; }
        a.rp8 = h->csr.rp8; a.col8 = h->csr.col8; a.a8 = h->csr.a8; a.b8 = h->csr.b8;
        fh_prof_begin(h, "spmm");
        fh_launch_spmm(a, ld, h->csr.is_complex != 0, h->csr.b_identity != 0, fh_spmm_grid(a.N, ld), h->stream);
        fh_prof_end(h);
        if (a.use_row_kernel) return fh_spmm_row_grid(a.N);
        if (lds_kernel) return (8 / (ld / 16)) * fh_spmm_lds_slots(a.nblk_rows, ld);
        return fh_spmm_partials(a.N, ld);     // partial-sum rows per node
    }
    fh_dense_op_args a;
    a.A = h->dense.A; a.B = h->dense.B; a.N = (int)h->dense.N; a.is_complex = h->dense.is_complex;
    // dense operator: complex128 panels only (fh_krylov forces prec 64 for dense matrices)
    a.nodes = c.nodes; a.X = (const cplx*)c.X; a.x_node_stride = c.x_stride; a.Y = (cplx*)c.Y; a.y_node_stride = c.y_stride;
    a.coefA = c.coefA; a.coefB = c.coefB; a.Bvec = (const cplx*)c.Bvec; a.b_node_stride = c.b_stride;
    a.U = (const cplx*)c.U; a.u_node_stride = c.u_stride; a.dot_mode = c.dot_mode;
    a.partial1 = c.partial1; a.partial2 = c.partial2; a.node_active = c.node_active;
    int nblk = fh_dense_op_nblk(a.N);
    fh_prof_begin(h, "dense_op");
    fh_launch_dense_op(a, ld, nblk, h->stream);
    fh_prof_end(h);
    return nblk;
}

static int fh_op_nblk(feasthip_ctx* h, int ld) {
    // (an upper bound is enough here: it sizes the partial-sum buffers; the row count used by the finalize kernels is what
    //  fh_apply_operator returns for the kernel it actually launched)
    if (h->kind == 2) return std::max(std::max(fh_spmm_partials((int)h->csr.N, ld), ld == 64 ? fh_spmm_row_grid((int)h->csr.N) : 0),
                                      h->csr.lcol ? (8 / (ld / 16)) * fh_spmm_lds_slots(h->csr.nblk, ld) : 0);
    return fh_dense_op_nblk((int)h->dense.N);
}
// row permutation of the panels (block order of a renumbered sparse matrix), or null
static const int* fh_perm(feasthip_ctx* h) { return h->kind == 2 ? h->csr.perm : nullptr; }
static int64_t fh_N(feasthip_ctx* h) { return h->kind == 2 ? h->csr.N : h->dense.N; }
static bool fh_b_identity(feasthip_ctx* h) { return h->kind == 2 ? h->csr.b_identity != 0 : h->dense.b_identity != 0; }

// wide = 1: the entry point also takes m > FH_MAX_LD (processed in 64-column panels)
static int fh_check_problem(feasthip_ctx* h, int64_t m, int wide = 0) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (h->poisoned) { h->last_error = "handle poisoned by an earlier device failure: destroy it"; return FEASTHIP_ERROR_INTERNAL; }
    if (h->kind == 0) { h->last_error = "no matrix set (feasthip_set_dense / feasthip_set_csr)"; return FEASTHIP_ERROR_N; }
    if (m <= 0 || (!wide && m > FH_MAX_LD) || m > fh_N(h)) {
        h->last_error = wide ? "block width m must satisfy 1 <= m <= N" : "block width m must satisfy 1 <= m <= min(N, 64)";
        return FEASTHIP_ERROR_M0;
    }
    return 0;
}

// upload per-column coefficient arrays [nodes][ld]
static int fh_upload_coefs(feasthip_ctx* h, const char* name, const std::vector<cplx>& host, cplx** dev) {
    void* p = nullptr;
    int rc = fh_get_buf(h, name, host.size() * sizeof(cplx), &p);
    if (rc) return rc;
    const size_t bytes = host.size() * sizeof(cplx);
    if (h->pin && bytes <= h->pin_cap / 4) {
        // staged through a pinned ring: the copy is queued and the caller's vector may go out of scope at once -- no
        // synchronisation per upload (a FEAST loop makes about twenty of these).  The ring wraps behind a stream
        // synchronisation, so a slot is never rewritten under a copy that is still queued.
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (h->pin_off + need > h->pin_cap) {
            FH_CHECK(hipStreamSynchronize(h->stream));
            h->pin_off = 0;
        }
        memcpy(h->pin + h->pin_off, host.data(), bytes);
        FH_CHECK(hipMemcpyAsync(p, h->pin + h->pin_off, bytes, hipMemcpyHostToDevice, h->stream));
        h->pin_off += need;
    } else {
        FH_CHECK(hipMemcpyAsync(p, host.data(), bytes, hipMemcpyHostToDevice, h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));   // host vector may go out of scope
    }
    *dev = (cplx*)p;
    return 0;
}

// Small host -> device copy through the pinned ring (queued, no synchronisation; falls back to a synchronous copy)
static int fh_upload_small(feasthip_ctx* h, void* dst, const void* src, size_t bytes) {
    if (h->pin && bytes <= h->pin_cap / 4) {
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (h->pin_off + need > h->pin_cap) { FH_CHECK(hipStreamSynchronize(h->stream)); h->pin_off = 0; }
        memcpy(h->pin + h->pin_off, src, bytes);
        FH_CHECK(hipMemcpyAsync(dst, h->pin + h->pin_off, bytes, hipMemcpyHostToDevice, h->stream));
        h->pin_off += need;
        return 0;
    }
    FH_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}
// Small device -> host copy: lands in the pinned ring (one DMA, no pageable staging), *slot points at it; valid after the
// caller's next stream synchronisation and until the ring wraps
static int fh_download_small(feasthip_ctx* h, const void* src, size_t bytes, const void** slot, std::vector<char>& fallback) {
    if (h->pin && bytes <= h->pin_cap / 4) {
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (h->pin_off + need > h->pin_cap) { FH_CHECK(hipStreamSynchronize(h->stream)); h->pin_off = 0; }
        FH_CHECK(hipMemcpyAsync(h->pin + h->pin_off, src, bytes, hipMemcpyDeviceToHost, h->stream));
        *slot = h->pin + h->pin_off;
        h->pin_off += need;
        return 0;
    }
    fallback.resize(bytes);
    FH_CHECK(hipMemcpyAsync(fallback.data(), src, bytes, hipMemcpyDeviceToHost, h->stream));
    *slot = fallback.data();
    return 0;
}

// ---------------------------------------------------------------------------------------
// batched Krylov solves on panels:  (z_e B - A) X_e = RHS  for e in [0, nodes)
//   method 0: BiCGStab (general), method 1: COCG (complex-symmetric S only)
//   prec 64 : everything in complex128; X holds the initial guess on entry.
//   prec 32 : mixed precision.  The fp64 residual r0 = RHS - S X0 of the initial guess is
//             normalised per column and narrowed to complex64; the correction S d = r0/||r0||
//             is solved in complex64 from a zero guess (all panels 8 B/element, reductions still
//             fp64) and added back, X = X0 + ||r0|| d.  The FEAST refinement loop only needs a
//             relative reduction of r0 (inexact solves), so single precision is ample; the
//             warm start, the residual and the Rayleigh-Ritz step stay fp64.
// ---------------------------------------------------------------------------------------
struct fh_solve_result {
    std::vector<int> node_iters, col_iters;
    int64_t iters_sum = 0;      // sum over nodes of max column iterations
    int64_t op_calls = 0;
    int max_iters = 0;
    std::vector<int> status;    // per node
    double max_rel_res = 0.0;
};

// sum_acc != null (COCG only): "sum mode" -- X keeps the initial guess, every step alpha p of every
// node is added, weighted with wnode[e], to the N x ld accumulator sum_acc (zeroed by the caller), so
// that  sum_e w_e X_e(final) = sum_e w_e X_e(initial) + sum_acc.
// shared_src != null (sum mode, prec 64): the initial residual of every node is  f_node,c * shared_src  with
// f = 1/(z_node - shared_lambda[c]) (device array, Ritz warm start) or 1 (shared_lambda == null: zero guess, the source
// is RHS).  X and RHS are then never read: no warm-start panels, no residual product, no separate P = R pass.
static int fh_krylov(feasthip_ctx* h, int method, int prec, int ld, int m, int nodes, const std::vector<cplx>& z,
                     const cplx* RHS, cplx* X, size_t stride, fh_solve_result& res, cplx* sum_acc = nullptr,
                     const std::vector<cplx>* wnode = nullptr, const cplx* shared_src = nullptr,
                     const double* shared_lambda = nullptr, const cplx* dznode = nullptr, const double* shared_lambda_host = nullptr) {
    if (h->kind != 2) prec = 64;          // the dense operator kernel takes complex128 panels only
    if (method != 1 || !wnode) sum_acc = nullptr;
    const int N = (int)fh_N(h);
    const size_t panel = (size_t)N * ld;
    if (stride != panel) { h->last_error = "internal: solution stride mismatch"; return FEASTHIP_ERROR_INTERNAL; }
    const size_t esz = prec == 32 ? sizeof(cplxf) : sizeof(cplx);
    int rc;
    void* p;
    // work panels R, Rhat, P, V, S, T (+ D and RHS32 for the mixed-precision correction)
    const int nvec = 6 + (prec == 32 ? 2 : 0);
    if ((rc = fh_get_buf(h, "kry_vecs", (size_t)nvec * nodes * panel * esz, &p))) return rc;
    char* base = (char*)p;
    auto vec = [&](int k) { return (void*)(base + (size_t)k * nodes * panel * esz); };
    void *R = vec(0), *Rh = vec(1), *P = vec(2), *V = vec(3), *S = vec(4), *T = vec(5);
    const size_t nl = (size_t)nodes * ld;
    if ((rc = fh_get_buf(h, "kry_scal_c", 4 * nl * sizeof(cplx), &p))) return rc;
    fh_krylov_scalars s;
    s.rho = (cplx*)p; s.alpha = s.rho + nl; s.omega = s.alpha + nl; s.beta = s.omega + nl;
    if ((rc = fh_get_buf(h, "kry_scal_d", 5 * nl * sizeof(double), &p))) return rc;
    s.r0norm = (double*)p; s.target = s.r0norm + nl; s.rnorm = s.target + nl;
    double* r0_64 = s.rnorm + nl;            // fp64 initial-residual norms (mixed precision)
    double* inv_r0 = r0_64 + nl;
    if ((rc = fh_get_buf(h, "kry_scal_i", (4 * nl + 2 * nodes + 4) * sizeof(int), &p))) return rc;
    s.active = (int*)p; s.iters = s.active + nl; s.status = s.iters + nl; s.node_active = s.status + nl;
    int* d_count = s.node_active + nodes;
    cplx* d_wnode = nullptr;
    // Fused COCG iteration (fh_sparse.hip): SpMM with five dots -> one finalize -> one vector kernel.  CSR operator through
    // the gather kernel only; FH_COCG_FUSED=0 selects the five-launch form for comparison.
    static const bool fused_off = getenv("FH_COCG_FUSED") && atoi(getenv("FH_COCG_FUSED")) == 0;
    const bool fused = method == 1 && h->kind == 2 && !fused_off;
    if (sum_acc || fused) {
        s.accum = d_count + 4; s.node_accum = s.accum + nl;
        FH_CHECK(hipMemsetAsync(s.accum, 0, (nl + nodes) * sizeof(int), h->stream));
    }
    if (sum_acc) {
        if ((rc = fh_upload_coefs(h, "kry_wnode", *wnode, &d_wnode))) return rc;
    }
    const int nblk_op = fh_op_nblk(h, ld);
    const int nblk_vec = fh_kry_nblk(N, ld, nodes);
    int fv_blk = 0, fv_seg = 0, fv_per = 0;
    int fv1_blk = 0, fv1_seg = 0, fv1_per = 0;        // geometry of the lazy start's first vector launch
    if (fused) fh_fused_vec_geometry(N, ld, prec == 32, &fv_blk, &fv_seg, &fv_per);
    if (fused) fh_fused_vec_geometry(N, ld, 1, &fv1_blk, &fv1_seg, &fv1_per);
    const int nblk_max = std::max(std::max(std::max(nblk_op, nblk_vec), fv_blk * fv_seg), fv1_blk * fv1_seg);
    if ((rc = fh_get_buf(h, "kry_partials", 2 * (size_t)nodes * nblk_max * ld * sizeof(cplx), &p))) return rc;
    cplx* part1 = (cplx*)p;
    cplx* part2 = part1 + (size_t)nodes * nblk_max * ld;
    cplx* sp[2] = {nullptr, nullptr};
    unsigned long long* d_tickets = nullptr;
    if (fused) {
        const size_t one = (size_t)nodes * nblk_op * ld;
        if ((rc = fh_get_buf(h, "kry_partials_op", 2 * one * sizeof(cplx), &p))) return rc;
        for (int q = 0; q < 2; ++q) sp[q] = (cplx*)p + q * one;
        if ((rc = fh_get_buf(h, "kry_tickets", (size_t)nodes * sizeof(unsigned long long), &p))) return rc;
        d_tickets = (unsigned long long*)p;
        FH_CHECK(hipMemsetAsync(d_tickets, 0, (size_t)nodes * sizeof(unsigned long long), h->stream));
    }

    // shifted-operator coefficients: S_e = z_e B - A
    std::vector<cplx> ca(nl), cb(nl);
    for (int e = 0; e < nodes; ++e)
        for (int c = 0; c < ld; ++c) { ca[e * ld + c] = cmake(-1, 0); cb[e * ld + c] = z[e]; }
    cplx *dca, *dcb;
    if ((rc = fh_upload_coefs(h, "kry_coefA", ca, &dca))) return rc;
    if ((rc = fh_upload_coefs(h, "kry_coefB", cb, &dcb))) return rc;

    fh_op_call oc;
    oc.m = m; oc.uniform_coef = 1; oc.coefA = dca; oc.coefB = dcb; oc.nodes = nodes;
    oc.partial1 = part1; oc.partial2 = part2; oc.U = nullptr; oc.u_stride = 0;
    fh_fin_args fa;
    fa.s = s; fa.partial1 = part1; fa.partial2 = part2; fa.m = m; fa.rtol = h->rtol; fa.atol = h->atol;
    fa.atol_scale = nullptr; fa.mode = method; fa.col_mask = nullptr;
    if (h->mask_live && !h->col_mask.empty()) {
        std::vector<int> mk(ld, 1);
        for (int c = 0; c < ld && c < (int)h->col_mask.size(); ++c) mk[c] = h->col_mask[c];
        if ((rc = fh_get_buf(h, "kry_colmask", ld * sizeof(int), &p))) return rc;
        FH_CHECK(hipMemcpyAsync(p, mk.data(), ld * sizeof(int), hipMemcpyHostToDevice, h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));
        fa.col_mask = (const int*)p;
    }
    fh_vec_args va;
    memset(&va, 0, sizeof(va));
    va.N = N; va.node_stride = panel; va.R = R; va.Rhat = Rh; va.P = P; va.V = V; va.S = S; va.T = T;
    va.s = s; va.partial1 = part1; va.partial2 = part2; va.prec = prec;
    va.counters = h->profiling ? h->d_counters : nullptr;
    va.sum_acc = sum_acc; va.wnode = d_wnode; va.sum_scale = (sum_acc && prec == 32) ? r0_64 : nullptr; va.nodes = nodes;

    void* Xk = X;        // the panel the Krylov recurrences update
    const bool shared_start = shared_src && sum_acc && method == 1 && prec == 64;
    if (shared_start) {
        // nothing to do here: R, P and the norms come from k_cocg_init_shared below
    } else if (prec == 64) {
        // R = RHS - S X0, ||R||^2
        oc.prec = 64; oc.X = X; oc.x_stride = panel; oc.Y = R; oc.y_stride = panel; oc.Bvec = RHS; oc.b_stride = 0;
        oc.dot_mode = 3; oc.node_active = nullptr;
        fa.nblk = fh_apply_operator(h, ld, oc);
        res.op_calls += 1;
    } else {
        // fp64 residual of the warm start into the (fp64-sized) tail of the work area
        void* q;
        if ((rc = fh_get_buf(h, "kry_r64", (size_t)nodes * panel * sizeof(cplx), &q))) return rc;
        cplx* R64 = (cplx*)q;
        oc.prec = 64; oc.X = X; oc.x_stride = panel; oc.Y = R64; oc.y_stride = panel; oc.Bvec = RHS; oc.b_stride = 0;
        oc.dot_mode = 3; oc.node_active = nullptr;
        int nb = fh_apply_operator(h, ld, oc);
        res.op_calls += 1;
        fh_fin_args f0 = fa;                       // only to obtain ||r0|| per column
        f0.nblk = nb; f0.rtol = 0.0; f0.atol = 0.0; f0.mode = 0;
        fh_launch_fin_init(f0, ld, nodes, h->stream);
        FH_CHECK(hipMemcpyAsync(r0_64, s.r0norm, nl * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        // narrow: RHS32 = R64 / ||r0||, D = 0, R = RHS32
        cplxf* D = (cplxf*)vec(6);
        cplxf* RHS32 = (cplxf*)vec(7);
        fh_launch_narrow_scaled(R64, panel, RHS32, panel, r0_64, N, ld, nblk_vec, nodes, h->stream);
        FH_CHECK(hipMemsetAsync(D, 0, (size_t)nodes * panel * sizeof(cplxf), h->stream));
        FH_CHECK(hipMemcpyAsync(R, RHS32, (size_t)nodes * panel * sizeof(cplxf), hipMemcpyDeviceToDevice, h->stream));
        Xk = D;
        // norms of the (unit) scaled residual for the stop test; atol is rescaled by 1/||r0||
        std::vector<double> hr0(nl), hinv(nl);
        FH_CHECK(hipMemcpyAsync(hr0.data(), r0_64, nl * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < nl; ++i) hinv[i] = hr0[i] > 0 ? 1.0 / hr0[i] : 0.0;
        FH_CHECK(hipMemcpyAsync(inv_r0, hinv.data(), nl * sizeof(double), hipMemcpyHostToDevice, h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));
        fa.atol_scale = inv_r0;
        // ||R||^2 of the narrowed residual through a zero-cost pass: reuse cocg_init/copy below
    }
    va.X = Xk;
    const int vprec = prec;
    // Lazy start (fused iteration over a CSR operator, both gather kernels): residual and direction of every node are the ONE source panel times a
    // per-node column factor, so neither is written here; the first operator product reads the source itself and the first
    // vector kernel writes R and P (fh_sparse.hip: k_cocg_init_lazy).  FH_NO_LAZY_START=1: materialise them as before.
    const bool lazy_off = getenv("FH_NO_LAZY_START") != nullptr;    // read per call (the tests flip it)
    const bool lazy = shared_start && fused && h->kind == 2 && !h->csr.is_complex && !lazy_off && (!shared_lambda || shared_lambda_host) &&
                      !(getenv("FH_LDS_SPMM") && atoi(getenv("FH_LDS_SPMM")) != 0);
    cplx* dfs = nullptr;
    if (lazy) {
        std::vector<cplx> fs(nl, cmake(1, 0));
        if (shared_lambda_host)
            for (int e = 0; e < nodes; ++e)
                for (int c = 0; c < m; ++c) fs[(size_t)e * ld + c] = cdiv(cmake(1, 0), cmake(z[e].x - shared_lambda_host[c], z[e].y));
        if ((rc = fh_upload_coefs(h, "kry_fscale", fs, &dfs))) return rc;
        fh_vec_args vs = va;
        vs.Q = shared_src; vs.first_scale = dfs;
        fh_launch_cocg_init_lazy(vs, ld, nblk_vec, nodes, h->stream);
        fa.nblk = nblk_vec;
        fh_launch_fin_init(fa, ld, nodes, h->stream);
    } else if (shared_start) {
        fh_vec_args vs = va;
        vs.Q = shared_src; vs.lambda = shared_lambda; vs.znode = dznode;
        fh_launch_cocg_init_shared(vs, ld, nblk_vec, nodes, h->stream);
        fa.nblk = nblk_vec;
        fh_launch_fin_init(fa, ld, nodes, h->stream);
    } else if (method == 1) {
        // COCG: P = R, rho = r^T r, ||r||
        fh_launch_cocg_init(va, ld, nblk_vec, nodes, h->stream);
        fa.nblk = nblk_vec;
        fh_launch_fin_init(fa, ld, nodes, h->stream);
    } else {
        if (prec == 32) {
            // partial2 = ||R||^2 of the narrowed residual (cocg_init also writes P = R)
            fh_launch_cocg_init(va, ld, nblk_vec, nodes, h->stream);
            fa.nblk = nblk_vec;
        }
        fa.mode = 0;
        fh_launch_fin_init(fa, ld, nodes, h->stream);
        fh_launch_copy_r(va, ld, nblk_vec, nodes, h->stream);            // Rhat = R ; P = R
    }
    (void)vprec;

    oc.prec = prec; oc.Bvec = nullptr; oc.b_stride = 0; oc.x_stride = panel; oc.y_stride = panel;
    // Iterate without ever blocking in the HIP runtime: chunks of `check_every` iterations are
    // queued back to back, each followed by a tiny kernel that publishes (chunk tag, active
    // columns) to a host-mapped word.  The host polls that word, stays at most two chunks ahead
    // of the device and stops queueing once a published count is zero.  (A hipStreamSynchronize
    // per chunk idled the GPU for milliseconds each time: 1.1 s -> 0.7 s per cfg-3 solve.)
    const int check_every = getenv("FH_CHECK_EVERY") ? std::max(1, atoi(getenv("FH_CHECK_EVERY"))) : 16;
    (void)d_count;
    *h->h_progress = 0ull;
    int it = 0;
    unsigned tag = 0;
    bool all_done = false;
    auto t_loop0 = std::chrono::steady_clock::now();
    auto progress = [&](unsigned& seen_tag, unsigned& seen_cnt) {
        unsigned long long w = *h->h_progress;
        seen_tag = (unsigned)(w >> 32); seen_cnt = (unsigned)(w & 0xffffffffull);
    };
    while (it < h->maxit && !all_done) {
        int chunk = std::min(check_every, h->maxit - it);
        for (int k = 0; k < chunk; ++k) {
            if (method == 0) {
                // V = S P, sigma = <Rhat, V>
                oc.X = P; oc.Y = V; oc.U = Rh; oc.u_stride = panel; oc.dot_mode = 1; oc.node_active = s.node_active;
                fa.nblk = fh_apply_operator(h, ld, oc);
                fh_prof_begin(h, "dot_finalize"); fh_launch_fin_alpha(fa, ld, nodes, h->stream); fh_prof_end(h);
                fh_prof_begin(h, "bicg_s"); fh_launch_s_update(va, ld, nblk_vec, nodes, h->stream); fh_prof_end(h);
                // T = S S, <T,S>, <T,T>
                oc.X = S; oc.Y = T; oc.U = nullptr; oc.dot_mode = 2;
                fa.nblk = fh_apply_operator(h, ld, oc);
                fh_prof_begin(h, "dot_finalize"); fh_launch_fin_omega(fa, ld, nodes, h->stream); fh_prof_end(h);
                fh_prof_begin(h, "bicg_xr"); fh_launch_xr_update(va, ld, nblk_vec, nodes, h->stream); fh_prof_end(h);
                fa.nblk = nblk_vec;
                fh_prof_begin(h, "dot_finalize"); fh_launch_fin_rho(fa, ld, nodes, h->stream); fh_prof_end(h);
                fh_prof_begin(h, "bicg_p"); fh_launch_p_update(va, ld, nblk_vec, nodes, h->stream); fh_prof_end(h);
                res.op_calls += 2;
            } else if (fused) {
                // Q = S P (stored in V), sigma = p^T q, kappa = q^T q (fh_sparse.hip, fused COCG)
                const bool first_lazy = lazy && it + k == 0;
                oc.X = first_lazy ? (const void*)shared_src : P; oc.x_stride = first_lazy ? 0 : panel; oc.colscale = first_lazy ? dfs : nullptr;
                oc.Y = V; oc.U = nullptr; oc.dot_mode = 6; oc.node_active = s.node_active;
                oc.partial1 = sp[0]; oc.partial2 = sp[1];
                va.first_src = first_lazy ? shared_src : nullptr; va.first_scale = first_lazy ? dfs : nullptr;
                fh_fused_fin_args ff;
                ff.s = s; ff.sig = sp[0]; ff.kap = sp[1];
                ff.rho = part1; ff.rr = part2; ff.tickets = d_tickets; ff.final_check = 0;
                ff.predict_stop = (h->rtol >= 1e-3 && h->atol == 0.0) ? 1 : 0;
                ff.nblk_op = fh_apply_operator(h, ld, oc);
                ff.nblk_vec = (it + k == 0) ? fa.nblk : ((lazy && it + k == 1) ? fv1_blk * fv1_seg : fv_blk * fv_seg);      // first iteration: the init kernel's partial rows; second (lazy start): the half-geometry launch's
                fh_prof_begin(h, "dot_finalize"); fh_launch_fused_fin(ff, ld, nodes, h->stream); fh_prof_end(h);
                fh_prof_begin(h, "cocg_vec"); fh_launch_fused_vec(va, ld, h->stream); fh_prof_end(h);
                res.op_calls += 1;
            } else {
                // Q = S P (stored in V), sigma = p^T S p
                oc.X = P; oc.Y = V; oc.U = nullptr; oc.dot_mode = 4; oc.node_active = s.node_active;
                fa.nblk = fh_apply_operator(h, ld, oc);
                fh_prof_begin(h, "dot_finalize"); fh_launch_fin_alpha(fa, ld, nodes, h->stream); fh_prof_end(h);
                fh_prof_begin(h, "cocg_xr"); fh_launch_cocg_update(va, ld, nblk_vec, nodes, h->stream); fh_prof_end(h);
                fa.nblk = nblk_vec;
                fh_prof_begin(h, "dot_finalize"); fh_launch_fin_rho(fa, ld, nodes, h->stream); fh_prof_end(h);
                fh_prof_begin(h, "cocg_p");
                if (sum_acc) fh_launch_cocg_p_sum(va, ld, nodes, h->stream);
                else fh_launch_cocg_p(va, ld, nblk_vec, nodes, h->stream);
                fh_prof_end(h);
                res.op_calls += 1;
            }
        }
        it += chunk;
        ++tag;
        fh_launch_publish_progress(s.node_active, nodes, h->d_progress, tag, h->stream);
        // throttle: wait (by polling host memory) until the device has finished chunk tag-2
        unsigned st = 0, sc = 0;
        for (unsigned spins = 1;; ++spins) {
            progress(st, sc);
            if (st >= 1 && sc == 0) { all_done = true; break; }
            if (tag < 3 || st + 3 > tag) break;      // at most three chunks queued ahead of the device
            std::this_thread::sleep_for(std::chrono::microseconds(100));   // poll, do not burn the core
            if ((spins & 2047u) == 0) {              // every ~0.2 s: a faulted queue never publishes; do not wait for it
                const hipError_t q = hipStreamQuery(h->stream);
                if (q != hipSuccess && q != hipErrorNotReady) {
                    h->last_error = std::string("device queue failed while iterating: ") + hipGetErrorString(q);
                    h->poisoned = 1;
                    return FEASTHIP_ERROR_INTERNAL;
                }
                // a wedged kernel keeps answering "not ready": overall deadline, generous against the slowest
                // measured iteration (2 ms with 16 nodes x 64 columns at N = 50 000), scaled by the problem size
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count();
                const double budget = 30.0 + 0.05 * (double)h->maxit * (1.0 + (double)N * nodes / 8.0e5);
                if (waited > budget) {
                    h->last_error = "device did not make progress on the Krylov iterations within the deadline (" + std::to_string((int)budget) + " s)";
                    fh_poison_unless_drained(h, 2.0);
                    return FEASTHIP_ERROR_INTERNAL;
                }
            }
        }
    }
    if (fused && it > 0) {
        // the stop test of the last step: true norms from the last vector kernel's partials (no SpMM follows it)
        fh_fused_fin_args ff;
        ff.s = s; ff.sig = ff.kap = nullptr; ff.rho = part1; ff.rr = part2; ff.tickets = d_tickets;
        ff.nblk_op = 0; ff.nblk_vec = (lazy && it == 1) ? fv1_blk * fv1_seg : fv_blk * fv_seg; ff.final_check = 1; ff.predict_stop = 0;
        fh_launch_fused_fin(ff, ld, nodes, h->stream);
    }
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (getenv("FH_DEBUG_TIMING"))
        fprintf(stderr, "[fh_krylov] nodes=%d its queued=%d loop wall %.3f ms\n", nodes, it,
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count());
    if (prec == 32 && !sum_acc)    // X = X0 + ||r0|| d
        fh_launch_widen_axpy(X, panel, (const cplxf*)Xk, panel, r0_64, N, ld, nblk_vec, nodes, h->stream);

    // gather per-column bookkeeping
    std::vector<int> iters(nl), status(nl), active(nl);
    std::vector<double> rnorm(nl), r0(nl);
    FH_CHECK(hipMemcpy(iters.data(), s.iters, nl * sizeof(int), hipMemcpyDeviceToHost));
    FH_CHECK(hipMemcpy(status.data(), s.status, nl * sizeof(int), hipMemcpyDeviceToHost));
    FH_CHECK(hipMemcpy(active.data(), s.active, nl * sizeof(int), hipMemcpyDeviceToHost));
    FH_CHECK(hipMemcpy(rnorm.data(), s.rnorm, nl * sizeof(double), hipMemcpyDeviceToHost));
    FH_CHECK(hipMemcpy(r0.data(), s.r0norm, nl * sizeof(double), hipMemcpyDeviceToHost));
    res.status.assign(nodes, 0);
    for (int e = 0; e < nodes; ++e) {
        int mx = 0, st = 0;
        for (int c = 0; c < m; ++c) {
            int i = e * ld + c;
            mx = std::max(mx, iters[i]);
            res.col_iters.push_back(iters[i]);
            if (active[i] || !std::isfinite(rnorm[i])) st = std::max(st, (int)FEASTHIP_ERROR_NO_CONVERGENCE);
            else if (status[i] == 8 && !(rnorm[i] <= h->atol + h->rtol * r0[i])) st = std::max(st, (int)FEASTHIP_ERROR_NO_CONVERGENCE);
            if (r0[i] > 0) res.max_rel_res = std::max(res.max_rel_res, rnorm[i] / r0[i]);
        }
        res.iters_sum += mx;
        res.node_iters.push_back(mx);
        res.max_iters = std::max(res.max_iters, mx);
        res.status[e] = st;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// restarted GMRES(m) on panels -- the reference's iterative solver
// (solve_shifted_iterative!, src/sparse/feast_sparse.jl:164-203; Krylov.jl gmres with
// restart=true, memory=m, zero initial guess unless X holds one, stop ||r|| <= atol + rtol ||r0||).
// Device resident (fh_gmres.hip): all columns of all local nodes advance in lock-step; basis panels, Hessenberg
// columns, Givens rotations and the per-column stop test live on the device.  The host only queues kernels: it
// looks at the device's published progress word without blocking to stop queueing once every column has converged
// inside a cycle, and reads the true-residual check once per restart cycle.
// ---------------------------------------------------------------------------------------
static int fh_gmres(feasthip_ctx* h, int ld, int m, int nodes_all, const std::vector<cplx>& z, const cplx* RHS, cplx* X,
                    size_t stride, fh_solve_result& res) {
    const int N = (int)fh_N(h);
    const size_t panel = (size_t)N * ld;
    const int mr = std::max(h->restart, 2);
    int rc;
    void* p;
    res.status.assign(nodes_all, 0);
    // node batches: the basis costs (mr + 2) panels per node
    const size_t per_node = (size_t)(mr + 2) * panel * sizeof(cplx);
    // budget: at most 48 GiB and at most half of what the device has free right now (plus what this handle already
    // holds for the basis) -- several ranks may share one card (shm rehearsal layout), and parts with less HBM exist
    size_t budget = (size_t)48 << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            auto held = h->bufs.find("gm_V");
            const size_t avail = free_b / 2 + (held != h->bufs.end() ? held->second.second : 0);
            budget = std::min(budget, std::max(avail, per_node));
        }
    }
    if (getenv("FH_GMRES_BUDGET_MB")) budget = (size_t)std::max(1, atoi(getenv("FH_GMRES_BUDGET_MB"))) << 20;
    int nbatch = (int)std::max<size_t>(1, std::min<size_t>((size_t)nodes_all, budget / std::max<size_t>(per_node, 1)));
    // an allocation failure halves the batch before it becomes an error
    for (;;) {
        if (fh_get_buf(h, "gm_V", (size_t)nbatch * (mr + 1) * panel * sizeof(cplx), &p) == 0 &&
            fh_get_buf(h, "gm_W", (size_t)nbatch * panel * sizeof(cplx), &p) == 0) break;
        if (nbatch == 1) return FEASTHIP_ERROR_MEMORY;
        hipGetLastError();                                   // clear the sticky out-of-memory status
        nbatch = (nbatch + 1) / 2;
    }
    const int nblk_vec = fh_kry_nblk(N, ld, nbatch);
    const int nblk_op = fh_op_nblk(h, ld);
    for (int e0 = 0; e0 < nodes_all; e0 += nbatch) {
        const int nodes = std::min(nbatch, nodes_all - e0);
        const size_t nl = (size_t)nodes * ld;
        if ((rc = fh_get_buf(h, "gm_V", (size_t)nodes * (mr + 1) * panel * sizeof(cplx), &p))) return rc;
        cplx* V = (cplx*)p;
        if ((rc = fh_get_buf(h, "gm_W", (size_t)nodes * panel * sizeof(cplx), &p))) return rc;
        cplx* W = (cplx*)p;
        if ((rc = fh_get_buf(h, "gm_part", fh_gm_partial_elems(mr, nblk_vec, nodes, ld) * sizeof(cplx), &p))) return rc;
        cplx* part = (cplx*)p;
        if ((rc = fh_get_buf(h, "gm_npart", (size_t)nodes * std::max(nblk_vec, nblk_op) * ld * sizeof(cplx), &p))) return rc;
        cplx* npart = (cplx*)p;
        const size_t hsz = nl * (size_t)(mr + 1) * mr;
        if ((rc = fh_get_buf(h, "gm_small_c", (hsz + nl * (mr + 1) * 2 + nl * mr * 3) * sizeof(cplx), &p))) return rc;
        cplx* sc = (cplx*)p;
        if ((rc = fh_get_buf(h, "gm_small_d", nl * 4 * sizeof(double), &p))) return rc;
        double* sd = (double*)p;
        if ((rc = fh_get_buf(h, "gm_small_i", (nl * 4 + nodes) * sizeof(int), &p))) return rc;
        int* si = (int*)p;
        fh_gmres_args ga;
        ga.N = N; ga.mr = mr; ga.panel = panel; ga.V = V; ga.v_node_stride = (size_t)(mr + 1) * panel; ga.W = W;
        ga.partial = part; ga.npartial = npart;
        ga.H = sc; ga.hcur = ga.H + hsz; ga.g = ga.hcur + nl * (mr + 1); ga.cs = ga.g + nl * (mr + 1); ga.sn = ga.cs + nl * mr;
        ga.y = ga.sn + nl * mr;
        ga.inv = sd; ga.r0norm = sd + nl; ga.target = sd + 2 * nl; ga.rnorm = sd + 3 * nl;
        ga.active = si; ga.iters = si + nl; ga.status = si + 2 * nl; ga.kdim = si + 3 * nl; ga.node_active = si + 4 * nl;
        FH_CHECK(hipMemsetAsync(sc, 0, hsz * sizeof(cplx), h->stream));

        std::vector<cplx> ca(nl, cmake(-1, 0)), cb(nl);
        for (int e = 0; e < nodes; ++e) for (int c = 0; c < ld; ++c) cb[(size_t)e * ld + c] = z[e0 + e];
        cplx *dca, *dcb;
        if ((rc = fh_upload_coefs(h, "gm_coefA", ca, &dca))) return rc;
        if ((rc = fh_upload_coefs(h, "gm_coefB", cb, &dcb))) return rc;
        cplx* Xb = X + (size_t)e0 * stride;
        fh_op_call oc;
        oc.m = m; oc.uniform_coef = 1; oc.coefA = dca; oc.coefB = dcb; oc.nodes = nodes; oc.prec = 64;
        oc.partial1 = nullptr; oc.partial2 = npart; oc.U = nullptr; oc.u_stride = 0;

        int total_it = 0, first = 1;
        unsigned tag = 0;
        *h->h_progress = 0ull;
        auto seen = [&](unsigned& st, unsigned& cnt) { unsigned long long w = *h->h_progress; st = (unsigned)(w >> 32); cnt = (unsigned)(w & 0xffffffffull); };
        while (true) {
            // true residual r = b - S x (into W), ||r|| per column, activity from the stop test
            oc.X = Xb; oc.x_stride = stride; oc.Y = W; oc.y_stride = panel; oc.Bvec = RHS; oc.b_stride = 0; oc.dot_mode = 3; oc.node_active = nullptr;
            const int nb_norm = fh_apply_operator(h, ld, oc);
            res.op_calls += 1;
            fh_launch_gm_start(ga, ld, nb_norm, nodes, first, h->rtol, h->atol, m, h->stream);
            first = 0;
            ++tag;
            fh_launch_publish_progress(ga.node_active, nodes, h->d_progress, tag, h->stream);
            FH_CHECK(hipStreamSynchronize(h->stream));           // the one host round trip per restart cycle
            unsigned st = 0, cnt = 0;
            seen(st, cnt);
            if (cnt == 0 || total_it >= h->maxit) break;
            fh_launch_gm_scale_store(ga, ld, W, panel, 0, nblk_vec, nodes, h->stream);           // v_0 = r / beta
            int ksteps = 0;
            const unsigned tag0 = tag;
            for (int k = 0; k < mr && total_it < h->maxit; ++k) {
                oc.X = V + (size_t)k * panel; oc.x_stride = ga.v_node_stride; oc.Y = W; oc.y_stride = panel; oc.Bvec = nullptr;
                oc.dot_mode = 0; oc.node_active = ga.node_active; oc.partial2 = nullptr;
                fh_apply_operator(h, ld, oc);                                                  // w = S v_k
                oc.partial2 = npart;
                res.op_calls += 1;
                fh_prof_begin(h, "gmres_ortho");
                fh_launch_gm_orthogonalize(ga, ld, k, nblk_vec, nodes, h->stream);             // CGS2 against v_0..v_k
                fh_launch_gm_givens(ga, ld, k, nblk_vec, nodes, h->stream);
                fh_launch_gm_scale_store(ga, ld, W, panel, k + 1, nblk_vec, nodes, h->stream); // v_{k+1} = w / h_{k+1,k}
                fh_prof_end(h);
                ++tag;
                fh_launch_publish_progress(ga.node_active, nodes, h->d_progress, tag, h->stream);
                ++ksteps; ++total_it;
                // non-blocking look at the device's progress: stop queueing steps once every column has converged
                seen(st, cnt);
                if (st > tag0 && cnt == 0) break;
                if (tag - st > 6) {                       // stay at most six steps ahead of the device
                    for (unsigned spins = 1; tag - st > 6; ++spins) {
                        std::this_thread::sleep_for(std::chrono::microseconds(100));
                        seen(st, cnt);
                        if ((spins & 2047u) == 0) {
                            const hipError_t q = hipStreamQuery(h->stream);
                            if (q != hipSuccess && q != hipErrorNotReady) {
                                h->last_error = std::string("device queue failed inside a GMRES cycle: ") + hipGetErrorString(q);
                                return FEASTHIP_ERROR_INTERNAL;
                            }
                        }
                    }
                    if (st > tag0 && cnt == 0) break;
                }
            }
            fh_launch_gm_finish_cycle(ga, ld, Xb, stride, ksteps, nblk_vec, nodes, h->stream);   // x += V y
        }
        // bookkeeping
        std::vector<int> iters(nl), status(nl), active(nl);
        std::vector<double> rnorm(nl), r0(nl), target(nl);
        FH_CHECK(hipMemcpy(iters.data(), ga.iters, nl * sizeof(int), hipMemcpyDeviceToHost));
        FH_CHECK(hipMemcpy(status.data(), ga.status, nl * sizeof(int), hipMemcpyDeviceToHost));
        FH_CHECK(hipMemcpy(active.data(), ga.active, nl * sizeof(int), hipMemcpyDeviceToHost));
        FH_CHECK(hipMemcpy(rnorm.data(), ga.rnorm, nl * sizeof(double), hipMemcpyDeviceToHost));
        FH_CHECK(hipMemcpy(r0.data(), ga.r0norm, nl * sizeof(double), hipMemcpyDeviceToHost));
        FH_CHECK(hipMemcpy(target.data(), ga.target, nl * sizeof(double), hipMemcpyDeviceToHost));
        for (int e = 0; e < nodes; ++e) {
            int mx = 0, stn = 0;
            for (int c = 0; c < m; ++c) {
                const size_t i = (size_t)e * ld + c;
                mx = std::max(mx, iters[i]);
                res.col_iters.push_back(iters[i]);
                if (active[i] || !std::isfinite(rnorm[i]) || rnorm[i] > target[i]) stn = FEASTHIP_ERROR_NO_CONVERGENCE;
                if (r0[i] > 0) res.max_rel_res = std::max(res.max_rel_res, rnorm[i] / r0[i]);
            }
            res.iters_sum += mx; res.node_iters.push_back(mx); res.max_iters = std::max(res.max_iters, mx);
            res.status[e0 + e] = stn;
        }
    }
    return 0;
}

static bool fh_is_complex_input(feasthip_ctx* h) { return h->kind == 2 ? h->csr.is_complex != 0 : h->dense.is_complex != 0; }

// ---------------------------------------------------------------------------------------
// contour sweep
// ---------------------------------------------------------------------------------------
// Mixed-precision dense solves (factor_precision = 32): complex64 LU factors, fp64 iterative refinement
//     Y <- Y + LU32^-1 (RHS - (z_e B - A) Y)
// until the relative residual of every column is below max(rtol, 1e-14) (rtol of feasthip_set_solver; a
// value >= 1 means no refinement at all: plain complex64 solves for the early, inexact FEAST loops), or
// stops improving; at most 8 steps.
// The residual is the fp64 dense operator kernel, so the result has fp64 accuracy as long as
// cond(z_e B - A) * eps32 < 1.  `single`: one shift through fh_dense_lu_solve_single (nodes == 1).
static int fh_dense_lu_refined(feasthip_ctx* h, int ld, int m, int nodes, const std::vector<cplx>& z, const cplx* Rhs, cplx* Y,
                               size_t panel, std::vector<int>& status, int64_t* nfact, bool single, double* worst_out, bool banded = false) {
    const int N = (int)fh_N(h);
    int rc;
    void* p;
    // banded: the same loop over the complex64 band factors of the sparse direct solver (residual = fp64 SpMM)
    auto solve = [&](const cplx* rhs, size_t rhs_stride, cplx* out, int64_t* nf) -> int {
        if (banded) {
            if (single) return fh_banded_solve_single(h, ld, m, z[0], rhs, out, &status[0], nf);
            return fh_banded_solve_nodes(h, ld, m, nodes, z, rhs, rhs_stride, out, panel, status, nf);
        }
        if (single) return fh_dense_lu_solve_single(h, ld, m, z[0], rhs, out, &status[0], nf);
        return fh_dense_lu_solve_nodes(h, ld, m, nodes, z, rhs, rhs_stride, out, panel, status, nf);
    };
    if ((rc = solve(Rhs, 0, Y, nfact))) return rc;
    const double tol = std::max(h->rtol, 1e-14);
    if (tol >= 1.0) { if (worst_out) *worst_out = 0.0; return 0; }
    if ((rc = fh_get_buf(h, "lr_R", (size_t)nodes * panel * sizeof(cplx), &p))) return rc;
    cplx* R = (cplx*)p;
    if ((rc = fh_get_buf(h, "lr_D", (size_t)nodes * panel * sizeof(cplx), &p))) return rc;
    cplx* D = (cplx*)p;
    if ((rc = fh_get_buf(h, "lr_part", (size_t)fh_vec_nblk(N, ld) * ld * sizeof(cplx), &p))) return rc;
    cplx* part = (cplx*)p;
    if ((rc = fh_get_buf(h, "lr_dots", (size_t)(nodes + 1) * ld * sizeof(cplx), &p))) return rc;
    cplx* ddots = (cplx*)p;
    const size_t nl = (size_t)nodes * ld;
    std::vector<cplx> ca(nl, cmake(-1, 0)), cb(nl), mone(ld, cmake(-1, 0));
    for (int e = 0; e < nodes; ++e) for (int c = 0; c < ld; ++c) cb[(size_t)e * ld + c] = z[e];
    cplx *dca, *dcb, *dmone;
    if ((rc = fh_upload_coefs(h, "lr_coefA", ca, &dca))) return rc;
    if ((rc = fh_upload_coefs(h, "lr_coefB", cb, &dcb))) return rc;
    if ((rc = fh_upload_coefs(h, "lr_mone", mone, &dmone))) return rc;
    std::vector<cplx> dots((size_t)(nodes + 1) * ld);
    fh_launch_dot_cols(Rhs, Rhs, N, ld, part, ddots + (size_t)nodes * ld, h->stream);
    double prev = 1e300, worst = 0.0;
    for (int it = 0; it < 8; ++it) {
        fh_op_call oc;
        oc.m = m; oc.uniform_coef = 1; oc.prec = 64;
        oc.X = Y; oc.x_stride = panel; oc.Y = R; oc.y_stride = panel; oc.coefA = dca; oc.coefB = dcb;
        oc.Bvec = Rhs; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
        oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = nodes;
        fh_apply_operator(h, ld, oc);                               // R = RHS - S Y
        for (int e = 0; e < nodes; ++e)
            fh_launch_dot_cols(R + (size_t)e * panel, R + (size_t)e * panel, N, ld, part, ddots + (size_t)e * ld, h->stream);
        FH_CHECK(hipMemcpyAsync(dots.data(), ddots, dots.size() * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));
        worst = 0.0;
        for (int e = 0; e < nodes; ++e) {
            if (status[e]) continue;                                 // singular factor: reported, not refined
            for (int c = 0; c < m; ++c) {
                const double b2 = dots[(size_t)nodes * ld + c].x, r2 = dots[(size_t)e * ld + c].x;
                if (b2 > 0.0) worst = std::max(worst, std::sqrt(r2 / b2));
            }
        }
        if (!(worst > tol) || !(worst < 0.5 * prev) || !std::isfinite(worst)) break;
        prev = worst;
        if ((rc = solve(R, panel, D, nullptr))) return rc;          // D = LU32^-1 R
        for (int e = 0; e < nodes; ++e)
            fh_launch_axpy_cols(Y + (size_t)e * panel, D + (size_t)e * panel, dmone, N, ld, h->stream);   // Y += D
    }
    if (worst_out) *worst_out = worst;
    return 0;
}

// Moment matrices of a sweep that is wider than one panel (variant B with M0 > 64): the panel call below holds the
// solution block Y_e of ITS columns only, so it adds the block column  w_e Q_all^H Y_e  (all rows, its columns) to the
// host accumulators; the caller uploads them once every panel is done.
struct fh_moment_ctx {
    const cplx* dQ_all = nullptr;       // N x m_all column-major device pointer (the whole subspace)
    int m_all = 0, col0 = 0;            // total width, first column of the current panel
    std::vector<cplx>* aq = nullptr;    // m_all x m_all column-major host accumulators
    std::vector<cplx>* sq = nullptr;
};

// Panels handed in / left behind in the kernels' own row-major layout (resident refinement loop): Qp replaces the import of a
// column-major dQ, eigres is A q - lambda B q for the Ritz values passed as ritz_lambda (the shared start residual: saves its
// product), out receives Q_proj as a panel (the export to a column-major dQproj is skipped when that pointer is null).
struct fh_panel_io {
    const cplx* Qp = nullptr;
    const cplx* eigres = nullptr;
    cplx* out = nullptr;
};

static int fh_contour_apply_panel(feasthip_ctx* h, int64_t m64, const cplx* dQ, const double* ritz_lambda,
                                  cplx* dQproj, cplx* dzAq, cplx* dzSq, int* node_status, feasthip_stats* stats,
                                  const fh_moment_ctx* mom = nullptr, const fh_panel_io* io = nullptr) {
    int rc = fh_check_problem(h, m64);
    if (rc) return rc;
    if (h->zne.empty()) { h->last_error = "no contour set"; return FEASTHIP_ERROR_FPM; }
    auto t0 = std::chrono::steady_clock::now();
    FH_CHECK(hipSetDevice(h->device));
    const int m = (int)m64, ld = fh_pick_ld(m), N = (int)fh_N(h);
    const int nodes = h->node_count;
    const size_t panel = (size_t)N * ld;
    void* p;
    cplx* Qp = nullptr;
    if (io && io->Qp) {
        Qp = const_cast<cplx*>(io->Qp);          // read only below (the right-hand side when B = I)
    } else {
        if ((rc = fh_get_buf(h, "ca_Qp", panel * sizeof(cplx), &p))) return rc;
        Qp = (cplx*)p;
        fh_launch_to_panel(dQ, N, N, m, Qp, ld, h->stream, fh_perm(h));
    }
    cplx* Outp = io ? io->out : nullptr;
    if (!Outp) {
        if ((rc = fh_get_buf(h, "ca_out", panel * sizeof(cplx), &p))) return rc;
        Outp = (cplx*)p;
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    if (nodes == 0) {
        if (io && io->out) FH_CHECK(hipMemsetAsync(io->out, 0, panel * sizeof(cplx), h->stream));
        if (dQproj) FH_CHECK(hipMemsetAsync(dQproj, 0, (size_t)N * m * sizeof(cplx), h->stream));
        if (dzAq) FH_CHECK(hipMemsetAsync(dzAq, 0, (size_t)m * m * sizeof(cplx), h->stream));
        if (dzSq) FH_CHECK(hipMemsetAsync(dzSq, 0, (size_t)m * m * sizeof(cplx), h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));
        return 0;
    }
    // rhs = B Q  (hoisted out of the node loop; the reference recomputes it per node,
    // src/dense/feast_dense.jl:184 -- it is loop invariant)
    cplx* Rhs = Qp;
    if (!fh_b_identity(h)) {
        if ((rc = fh_get_buf(h, "ca_rhs", panel * sizeof(cplx), &p))) return rc;
        Rhs = (cplx*)p;
        std::vector<cplx> ca(ld, cmake(0, 0)), cb(ld, cmake(1, 0));
        cplx *dca, *dcb;
        if ((rc = fh_upload_coefs(h, "ca_coefA", ca, &dca))) return rc;
        if ((rc = fh_upload_coefs(h, "ca_coefB", cb, &dcb))) return rc;
        fh_op_call oc;
        oc.m = m;
        oc.X = Qp; oc.x_stride = 0; oc.Y = Rhs; oc.y_stride = 0; oc.coefA = dca; oc.coefB = dcb;
        oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
        oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
        fh_apply_operator(h, ld, oc);
    }
    std::vector<cplx> z(nodes), w(nodes);
    for (int e = 0; e < nodes; ++e) {
        z[e] = h->zne[h->node_ids[e]];
        w[e] = cscale(h->wne[h->node_ids[e]], h->weight_scale);
    }
    if ((rc = fh_get_buf(h, "ca_Y", (size_t)nodes * panel * sizeof(cplx), &p))) return rc;
    cplx* Y = (cplx*)p;
    std::vector<int> status(nodes, 0);
    cplx* sum_acc = nullptr;
    bool sum_shared = false;          // sum mode started from one shared residual panel: no per-node solution panels exist

    // destroyed on every return path (the solver branches below return early on errors)
    struct ev_guard { hipEvent_t a = nullptr, b = nullptr; ~ev_guard() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); } } evg;
    FH_CHECK(hipEventCreate(&evg.a)); FH_CHECK(hipEventCreate(&evg.b));
    const hipEvent_t ev0 = evg.a, ev1 = evg.b;
    FH_CHECK(hipEventRecord(ev0, h->stream));
    if (h->solver == FEASTHIP_SOLVER_LU) {
        if (h->kind != 1) { h->last_error = "solver LU requires a dense matrix (sparse direct factorisation is not provided; use BICGSTAB)"; return FEASTHIP_ERROR_FPM; }
        int64_t nfact = 0;
        double worst = 0.0;
        if (h->factor_precision == 32) rc = fh_dense_lu_refined(h, ld, m, nodes, z, Rhs, Y, panel, status, &nfact, false, &worst);
        else rc = fh_dense_lu_solve_nodes(h, ld, m, nodes, z, Rhs, 0, Y, panel, status, &nfact);
        if (rc) return rc;
        if (stats) { stats->factorizations = nfact; stats->max_rel_residual = worst; }
    } else if (h->solver == FEASTHIP_SOLVER_BANDED) {
        int64_t nfact = 0;
        double worst = 0.0;
        if (h->factor_precision == 32) rc = fh_dense_lu_refined(h, ld, m, nodes, z, Rhs, Y, panel, status, &nfact, false, &worst, true);
        else rc = fh_banded_solve_nodes(h, ld, m, nodes, z, Rhs, 0, Y, panel, status, &nfact);
        if (rc) return rc;
        if (stats) { stats->factorizations = nfact; stats->max_rel_residual = worst; }
    } else if (h->solver == FEASTHIP_SOLVER_BICGSTAB || h->solver == FEASTHIP_SOLVER_COCG) {
        if (h->solver == FEASTHIP_SOLVER_COCG && fh_is_complex_input(h)) {
            h->last_error = "solver COCG needs a complex-SYMMETRIC shifted matrix: real-symmetric A and B only";
            return FEASTHIP_ERROR_FPM;
        }
        // initial guess
        cplx* dz;
        if ((rc = fh_upload_coefs(h, "ca_z", z, &dz))) return rc;
        double* dlam = nullptr;
        if (ritz_lambda) {
            std::vector<double> lam(ld, 0.0);
            for (int c = 0; c < m; ++c) lam[c] = ritz_lambda[c];
            if ((rc = fh_get_buf(h, "ca_lam", ld * sizeof(double), &p))) return rc;
            dlam = (double*)p;
            FH_CHECK(hipMemcpy(dlam, lam.data(), ld * sizeof(double), hipMemcpyHostToDevice));
        }
        fh_solve_result sr;
        // sum mode: only Q_proj is wanted (no moments), so the per-node solutions are never formed
        if (h->solver == FEASTHIP_SOLVER_COCG && !dzAq && !dzSq && !mom && h->sum_mode) {
            if ((rc = fh_get_buf(h, "ca_acc", panel * sizeof(cplx), &p))) return rc;
            sum_acc = (cplx*)p;
            FH_CHECK(hipMemsetAsync(sum_acc, 0, panel * sizeof(cplx), h->stream));
        }
        // Shared start (sum mode, fp64 panels): the warm start Y0_e = q_c/(z_e - lambda_c) has the residual
        // (A q_c - lambda_c B q_c)/(z_e - lambda_c) -- ONE eigen-residual panel serves every node -- and its weighted
        // sum over the nodes is q_c * sum_e w_e/(z_e - lambda_c): neither the warm-start panels nor their residual
        // products are ever formed.  Zero guess: the residual is RHS for every node.
        const cplx* shared_src = nullptr;
        sum_shared = sum_acc && h->factor_precision == 64 && !getenv("FH_NO_SHARED_START");
        if (sum_shared) {
            if (ritz_lambda && io && io->eigres) {
                shared_src = io->eigres;                         // left behind by the Ritz step of the previous loop
            } else if (ritz_lambda) {
                std::vector<cplx> ca(ld, cmake(1, 0)), cb(ld, cmake(0, 0));
                for (int c = 0; c < m; ++c) cb[c] = cmake(-ritz_lambda[c], 0);
                cplx *dca, *dcb;
                if ((rc = fh_upload_coefs(h, "ca_rcoefA", ca, &dca))) return rc;
                if ((rc = fh_upload_coefs(h, "ca_rcoefB", cb, &dcb))) return rc;
                if ((rc = fh_get_buf(h, "ca_eigres", panel * sizeof(cplx), &p))) return rc;
                fh_op_call oc;
                oc.m = m; oc.uniform_coef = 0;
                oc.X = Qp; oc.x_stride = 0; oc.Y = p; oc.y_stride = 0; oc.coefA = dca; oc.coefB = dcb;
                oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
                oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
                fh_apply_operator(h, ld, oc);                    // A q - lambda B q (B = I handled by the operator kernel)
                shared_src = (const cplx*)p;
            } else {
                shared_src = Rhs;
            }
        } else {
            fh_vec_args va;
            memset(&va, 0, sizeof(va));
            va.N = N; va.node_stride = panel; va.X = Y; va.Q = Qp; va.lambda = dlam; va.znode = dz;
            fh_launch_init_guess(va, ld, fh_vec_nblk(N, ld), nodes, h->stream);
        }
        rc = fh_krylov(h, h->solver == FEASTHIP_SOLVER_COCG ? 1 : 0, h->factor_precision, ld, m, nodes, z, Rhs, Y, panel, sr,
                       sum_acc, &w, shared_src, dlam, dz, ritz_lambda);
        if (rc) return rc;
        status = sr.status;
        h->last_node_iters = sr.node_iters;
        h->last_col_iters = sr.col_iters;
        h->last_col_m = m;
        if (stats) {
            stats->krylov_iterations = sr.iters_sum;
            stats->spmm_calls = sr.op_calls;
            stats->max_rel_residual = sr.max_rel_res;
        }
    } else {
        // GMRES: zero initial guess like Krylov.jl (or the Ritz warm start when given)
        cplx* dz;
        if ((rc = fh_upload_coefs(h, "ca_z", z, &dz))) return rc;
        double* dlam = nullptr;
        if (ritz_lambda) {
            std::vector<double> lam(ld, 0.0);
            for (int c = 0; c < m; ++c) lam[c] = ritz_lambda[c];
            if ((rc = fh_get_buf(h, "ca_lam", ld * sizeof(double), &p))) return rc;
            dlam = (double*)p;
            FH_CHECK(hipMemcpy(dlam, lam.data(), ld * sizeof(double), hipMemcpyHostToDevice));
        }
        fh_vec_args va;
        memset(&va, 0, sizeof(va));
        va.N = N; va.node_stride = panel; va.X = Y; va.Q = Qp; va.lambda = dlam; va.znode = dz; va.prec = 64;
        fh_launch_init_guess(va, ld, fh_vec_nblk(N, ld), nodes, h->stream);
        fh_solve_result sr;
        rc = fh_gmres(h, ld, m, nodes, z, Rhs, Y, panel, sr);
        if (rc) return rc;
        status = sr.status;
        h->last_node_iters = sr.node_iters;
        h->last_col_iters = sr.col_iters;
        h->last_col_m = m;
        if (stats) {
            stats->krylov_iterations = sr.iters_sum;
            stats->spmm_calls = sr.op_calls;
            stats->max_rel_residual = sr.max_rel_res;
        }
    }
    FH_CHECK(hipEventRecord(ev1, h->stream));

    // Q_proj = sum_e (scale*w_e) Y_e
    cplx* dw;
    if ((rc = fh_upload_coefs(h, "ca_w", w, &dw))) return rc;
    fh_prof_begin(h, "accumulate");
    if (sum_shared) {
        cplx* drho = nullptr;
        if (ritz_lambda) {
            std::vector<cplx> rho(ld, cmake(0, 0));
            for (int c = 0; c < m; ++c)
                for (int e = 0; e < nodes; ++e) rho[c] = cadd(rho[c], cdiv(w[e], cmake(z[e].x - ritz_lambda[c], z[e].y)));
            if ((rc = fh_upload_coefs(h, "ca_rho", rho, &drho))) return rc;
        }
        fh_launch_sum_finish(Qp, drho, sum_acc, Outp, N, ld, h->real_projection, h->stream);
    } else {
        fh_launch_accumulate(Y, panel, dw, nodes, N, ld, sum_acc, Outp, h->real_projection, h->stream);
    }
    fh_prof_end(h);
    if (dQproj) fh_launch_from_panel(Outp, ld, N, m, dQproj, N, h->stream, fh_perm(h));

    // optional moments (variant B): zAq += w_e Q^H Y_e ; zSq += w_e z_e Q^H Y_e
    if (dzAq || dzSq || mom) {
        if ((rc = fh_get_buf(h, "gram_work", fh_gram_work_elems(ld) * sizeof(cplx), &p))) return rc;
        cplx* gw = (cplx*)p;
        if ((rc = fh_get_buf(h, "gram_G", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
        cplx* G = (cplx*)p;
        const int m_all = mom ? mom->m_all : m, col0 = mom ? mom->col0 : 0;
        std::vector<cplx> Gh((size_t)ld * ld), aq_local, sq_local;
        if (!mom) { aq_local.assign((size_t)m * m, cmake(0, 0)); sq_local.assign((size_t)m * m, cmake(0, 0)); }
        std::vector<cplx>& aq = mom ? *mom->aq : aq_local;
        std::vector<cplx>& sq = mom ? *mom->sq : sq_local;
        cplx* Qrow = Qp;                       // row block of Q in panel layout (the panel's own columns when not wide)
        if (mom && (rc = fh_get_buf(h, "ca_Qrow", panel * sizeof(cplx), &p))) return rc;
        if (mom) Qrow = (cplx*)p;
        for (int r0 = 0; r0 < m_all; r0 += ld) {
            const int mr_ = std::min(ld, m_all - r0);
            if (mom) {
                fh_launch_to_panel(mom->dQ_all + (size_t)r0 * N, N, N, mr_, Qrow, ld, h->stream, fh_perm(h));
            }
            for (int e = 0; e < nodes; ++e) {
                fh_prof_begin(h, "gram");
                fh_launch_gram(Qrow, Y + (size_t)e * panel, N, ld, 0, gw, G, h->stream);
                fh_prof_end(h);
                FH_CHECK(hipMemcpyAsync(Gh.data(), G, Gh.size() * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
                FH_CHECK(hipStreamSynchronize(h->stream));
                cplx wz = cmul(w[e], z[e]);
                for (int c2 = 0; c2 < m; ++c2)
                    for (int c1 = 0; c1 < mr_; ++c1) {
                        cplx g = Gh[(size_t)c2 * ld + c1];
                        cfma(aq[(size_t)(col0 + c2) * m_all + r0 + c1], w[e], g);
                        cfma(sq[(size_t)(col0 + c2) * m_all + r0 + c1], wz, g);
                    }
            }
            if (!mom) break;
        }
        if (!mom) {
            if (h->real_projection) for (size_t i = 0; i < aq.size(); ++i) { aq[i].y = 0.0; sq[i].y = 0.0; }
            if (dzAq) FH_CHECK(hipMemcpyAsync(dzAq, aq.data(), aq.size() * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
            if (dzSq) FH_CHECK(hipMemcpyAsync(dzSq, sq.data(), sq.size() * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
        }
    }
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (node_status) for (int e = 0; e < nodes; ++e) node_status[e] = status[e];
    if (stats) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev0, ev1);
        stats->seconds_solve = ms * 1e-3;
        stats->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    fh_prof_collect(h);
    FH_CHECK(hipGetLastError());       // launch-configuration errors do not surface through the stream sync
    return 0;
}

// m > 64: the columns of Q are independent right-hand sides, so the sweep runs panel by panel
// (64 columns each); LU factors are shared by the panels through the per-node cache.
static int fh_contour_apply_local(feasthip_ctx* h, int64_t m64, const cplx* dQ, const double* ritz_lambda,
                                  cplx* dQproj, cplx* dzAq, cplx* dzSq, int* node_status, feasthip_stats* stats) {
    if (m64 <= FH_MAX_LD) return fh_contour_apply_panel(h, m64, dQ, ritz_lambda, dQproj, dzAq, dzSq, node_status, stats);
    int rc = fh_check_problem(h, m64, 1);
    if (rc) return rc;
    const int m = (int)m64, N = (int)fh_N(h), nodes = h->node_count;
    // moment matrices of a wide sweep: block columns gathered on the host, uploaded after the last panel
    std::vector<cplx> aq_all, sq_all;
    fh_moment_ctx mom;
    const bool want_mom = dzAq || dzSq;
    if (want_mom) {
        aq_all.assign((size_t)m * m, cmake(0, 0)); sq_all.assign((size_t)m * m, cmake(0, 0));
        mom.dQ_all = dQ; mom.m_all = m; mom.aq = &aq_all; mom.sq = &sq_all;
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    std::vector<int> ns(std::max(nodes, 1), 0), node_it(nodes, 0), col_it((size_t)nodes * m, 0);
    if (node_status) for (int e = 0; e < nodes; ++e) node_status[e] = 0;
    const std::vector<int> mask = h->col_mask;
    for (int c0 = 0; c0 < m; c0 += FH_MAX_LD) {
        const int mc = std::min(FH_MAX_LD, m - c0);
        feasthip_stats st;
        h->col_mask.clear();
        if (!mask.empty()) for (int c = c0; c < c0 + mc; ++c) h->col_mask.push_back(c < (int)mask.size() ? mask[c] : 1);
        h->last_node_iters.clear(); h->last_col_iters.clear();
        mom.col0 = c0;
        rc = fh_contour_apply_panel(h, mc, dQ + (size_t)c0 * N, ritz_lambda ? ritz_lambda + c0 : nullptr,
                                    dQproj + (size_t)c0 * N, nullptr, nullptr, ns.data(), &st, want_mom ? &mom : nullptr);
        h->col_mask = mask;
        if (rc) return rc;
        for (int e = 0; e < nodes; ++e) {
            if (node_status) node_status[e] = std::max(node_status[e], ns[e]);
            if (e < (int)h->last_node_iters.size()) node_it[e] = std::max(node_it[e], h->last_node_iters[e]);
            for (int c = 0; c < mc; ++c)
                if ((size_t)e * mc + c < h->last_col_iters.size()) col_it[(size_t)e * m + c0 + c] = h->last_col_iters[(size_t)e * mc + c];
        }
        if (stats) {
            stats->seconds_total += st.seconds_total; stats->seconds_solve += st.seconds_solve;
            stats->krylov_iterations += st.krylov_iterations; stats->spmm_calls += st.spmm_calls;
            stats->factorizations += st.factorizations;
            stats->max_rel_residual = std::max(stats->max_rel_residual, st.max_rel_residual);
        }
    }
    h->last_node_iters = node_it; h->last_col_iters = col_it; h->last_col_m = m;
    if (want_mom) {
        if (h->real_projection) for (size_t i = 0; i < aq_all.size(); ++i) { aq_all[i].y = 0.0; sq_all[i].y = 0.0; }
        if (dzAq) FH_CHECK(hipMemcpy(dzAq, aq_all.data(), aq_all.size() * sizeof(cplx), hipMemcpyHostToDevice));
        if (dzSq) FH_CHECK(hipMemcpy(dzSq, sq_all.data(), sq_all.size() * sizeof(cplx), hipMemcpyHostToDevice));
    }
    return 0;
}

// The sweep as the caller sees it: this rank's nodes x this rank's column block (feasthip_set_column_block),
// then -- when a communicator is attached -- ONE packed all-reduce
//     [ Q_proj (N*m reals for a real-projection sweep, else 2*N*m) | zAq | zSq | per-node failure flags ]
// on the handle's stream: the image of MPI.Allreduce in src/parallel/feast_mpi.jl:117-119, 856-858 and of the
// master sum src/parallel/feast_parallel.jl:497-503.  With a communicator node_status is GLOBAL (ne entries,
// indexed by contour node), without one it is per local node as before.
// Resident form (rs != null; m <= 64, no moments): the subspace comes in as the panel rs->Q (with, optionally, its
// eigen-residual panel rs->eigres), the summed Q_proj is left in the panel rs->P (N x rs->ld); column blocks are cut out of
// / packed into the panels by fh_launch_panel_cols / fh_launch_pack_cols, and the reduce carries N x ld values.
struct fh_resident_sweep {
    const cplx* Q; const cplx* eigres; cplx* P; int ld;
};

static int fh_contour_apply_impl(feasthip_ctx* h, int64_t m64, const cplx* dQ, const double* ritz_lambda,
                                 cplx* dQproj, cplx* dzAq, cplx* dzSq, int* node_status, feasthip_stats* stats,
                                 const fh_resident_sweep* rs = nullptr) {
    const int nr = fh_comm_nranks(h);
    int64_t c0 = 0, c1 = m64;
    if (h->col_block_hi >= 0) { c0 = std::min(h->col_block_lo, m64); c1 = std::min(std::max(h->col_block_hi, c0), m64); }
    const bool full = (c0 == 0 && c1 == m64);
    if (nr == 1 && full) {
        if (rs) {
            fh_panel_io io; io.Qp = rs->Q; io.eigres = rs->eigres; io.out = rs->P;
            return fh_contour_apply_panel(h, m64, nullptr, ritz_lambda, nullptr, nullptr, nullptr, node_status, stats, nullptr, &io);
        }
        return fh_contour_apply_local(h, m64, dQ, ritz_lambda, dQproj, dzAq, dzSq, node_status, stats);
    }
    // The shape of the packed reduce depends only on what every rank was called with (N, m, ne, the moment pointers, the
    // projection mode).  An argument error is therefore the same on every rank and may return at once; anything that
    // can fail on ONE rank only (allocations, the sweep, copies) is recorded in local_rc and the rank still joins the
    // reduce with a zeroed payload and its failure flag set -- its peers are waiting in the collective, and RCCL has no
    // timeout.
    int rc = fh_check_problem(h, m64, rs ? 0 : 1);
    if (rc) return rc;
    if (!full && (dzAq || dzSq)) { h->last_error = "contour_apply: moment matrices need the full column block"; return FEASTHIP_ERROR_M0; }
    const int N = (int)fh_N(h), m = (int)m64, nodes = h->node_count, ne = (int)h->zne.size();
    std::vector<int> ns(std::max(nodes, 1), 0);
    if (stats) memset(stats, 0, sizeof(*stats));
    int local_rc = 0;
    auto soft = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && !local_rc) {
            h->last_error = std::string(what) + ": " + hipGetErrorString(e);
            local_rc = (e == hipErrorOutOfMemory) ? FEASTHIP_ERROR_MEMORY : FEASTHIP_ERROR_INTERNAL;
        }
    };
    soft(hipSetDevice(h->device), "hipSetDevice");
    // the reduce buffer comes first: without it this rank cannot join the collective at all
    const size_t nq = (size_t)N * (rs ? rs->ld : m) * (h->real_projection ? 1 : 2);
    const size_t nm = (size_t)m * m * 2;
    const size_t total = nq + (dzAq ? nm : 0) + (dzSq ? nm : 0) + 3 * (size_t)ne + 1;
    double* pack = nullptr;
    if (nr > 1) {
        void* p = nullptr;
        if ((rc = fh_get_buf(h, "comm_pack", total * sizeof(double), &p))) {
            // nothing to reduce into: tell the peers through the transport's own failure path where there is one
            // (shm: the failed flag releases their barriers), then give up -- the caller must treat this as fatal
            fh_comm_mark_failed(h);
            return rc;
        }
        pack = (double*)p;
    }
    if (!local_rc && !full && !rs) soft(hipMemsetAsync(dQproj, 0, (size_t)N * m * sizeof(cplx), h->stream), "hipMemsetAsync(Q_proj)");
    cplx* rs_out = nullptr;                       // resident form: this rank's block of Q_proj, an N x rs_ldw panel
    int rs_ldw = 0;
    if (rs && !local_rc && c1 > c0) {
        const int w = (int)(c1 - c0);
        rs_ldw = fh_pick_ld(w);
        fh_panel_io io;
        void* p = nullptr;
        int brc = fh_get_buf(h, "rs_Osub", (size_t)N * rs_ldw * sizeof(cplx), &p);
        rs_out = (cplx*)p;
        if (!brc && full) { io.Qp = rs->Q; io.eigres = rs->eigres; }
        if (!brc && !full) {
            // this rank's columns of the subspace (and of its eigen-residual) as panels of their own
            if (!(brc = fh_get_buf(h, "rs_Qsub", (size_t)N * rs_ldw * sizeof(cplx), &p))) {
                fh_launch_panel_cols(rs->Q, rs->ld, (int)c0, w, N, (cplx*)p, rs_ldw, h->stream);
                io.Qp = (const cplx*)p;
            }
            if (!brc && rs->eigres && !(brc = fh_get_buf(h, "rs_Esub", (size_t)N * rs_ldw * sizeof(cplx), &p))) {
                fh_launch_panel_cols(rs->eigres, rs->ld, (int)c0, w, N, (cplx*)p, rs_ldw, h->stream);
                io.eigres = (const cplx*)p;
            }
        }
        if (brc) local_rc = brc;
        else {
            io.out = rs_out;
            const std::vector<int> mask = h->col_mask;
            if (!mask.empty()) {
                h->col_mask.clear();
                for (int64_t c = c0; c < c1; ++c) h->col_mask.push_back(c < (int64_t)mask.size() ? mask[c] : 1);
            }
            local_rc = fh_contour_apply_panel(h, w, nullptr, ritz_lambda ? ritz_lambda + c0 : nullptr, nullptr, nullptr, nullptr,
                                              ns.data(), stats, nullptr, &io);
            h->col_mask = mask;
        }
    } else if (rs && !local_rc) {
        soft(hipStreamSynchronize(h->stream), "hipStreamSynchronize");
    } else if (!local_rc && c1 > c0) {
        const std::vector<int> mask = h->col_mask;
        if (!mask.empty()) {
            h->col_mask.clear();
            for (int64_t c = c0; c < c1; ++c) h->col_mask.push_back(c < (int64_t)mask.size() ? mask[c] : 1);
        }
        local_rc = fh_contour_apply_local(h, c1 - c0, dQ + (size_t)c0 * N, ritz_lambda ? ritz_lambda + c0 : nullptr,
                                          dQproj + (size_t)c0 * N, dzAq, dzSq, ns.data(), stats);
        h->col_mask = mask;
    } else if (!local_rc) {
        soft(hipStreamSynchronize(h->stream), "hipStreamSynchronize");
    }
    if (nr == 1) {
        if (local_rc) return local_rc;
        if (rs) {                                 // one rank, a column block: the other columns of Q_proj are zero
            FH_CHECK(hipMemsetAsync(rs->P, 0, (size_t)N * rs->ld * sizeof(cplx), h->stream));
            if (rs_out) {
                void* p = nullptr;
                if ((rc = fh_get_buf(h, "comm_pack", nq * sizeof(double), &p))) return rc;
                fh_launch_pack_cols(rs_out, rs_ldw, (int)c0, (int)(c1 - c0), N, (double*)p, rs->ld, h->real_projection, h->stream);
                if (h->real_projection) fh_launch_unpack_real((const double*)p, rs->P, nq, h->stream);
                else FH_CHECK(hipMemcpyAsync(rs->P, p, nq * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            }
            FH_CHECK(hipStreamSynchronize(h->stream));
        }
        if (node_status) for (int e = 0; e < nodes; ++e) node_status[e] = ns[e];
        return 0;
    }
    size_t off = nq;
    if (!local_rc) {
        if (rs && rs_out) fh_launch_pack_cols(rs_out, rs_ldw, (int)c0, (int)(c1 - c0), N, pack, rs->ld, h->real_projection, h->stream);
        else if (rs) soft(hipMemsetAsync(pack, 0, nq * sizeof(double), h->stream), "pack Q_proj (no columns)");
        else if (h->real_projection) fh_launch_pack_real(dQproj, pack, nq, h->stream);
        else soft(hipMemcpyAsync(pack, dQproj, nq * sizeof(double), hipMemcpyDeviceToDevice, h->stream), "pack Q_proj");
        if (dzAq) { soft(hipMemcpyAsync(pack + off, dzAq, nm * sizeof(double), hipMemcpyDeviceToDevice, h->stream), "pack zAq"); off += nm; }
        if (dzSq) { soft(hipMemcpyAsync(pack + off, dzSq, nm * sizeof(double), hipMemcpyDeviceToDevice, h->stream), "pack zSq"); off += nm; }
    } else {
        off += (dzAq ? nm : 0) + (dzSq ? nm : 0);
    }
    // tail of the packed buffer: [no-convergence flags | singular flags | Krylov iterations per contour node | rank failed]
    std::vector<double> flags(3 * (size_t)ne + 1, 0.0);
    for (int e = 0; e < nodes && !local_rc; ++e) {
        const int g = h->node_ids[e];
        if (ns[e] == FEASTHIP_ERROR_LAPACK) flags[ne + g] = 1.0;
        else if (ns[e] != 0) flags[g] = 1.0;
        if (c1 > c0 && e < (int)h->last_node_iters.size()) flags[2 * (size_t)ne + g] = (double)h->last_node_iters[e];
    }
    hipError_t e_flags = hipSuccess;
    if (local_rc) {
        // our own payload may be garbage: contribute zeros so that the peers' sums stay finite
        e_flags = hipMemsetAsync(pack, 0, off * sizeof(double), h->stream);
    }
    flags[3 * (size_t)ne] = local_rc ? 1.0 : 0.0;
    if (e_flags == hipSuccess) e_flags = hipMemcpyAsync(pack + off, flags.data(), flags.size() * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e_flags != hipSuccess) {
        // the stream itself refuses work: the flag cannot be shipped.  Still enter the collective (whatever the buffer
        // holds) so that the peers return; this rank reports the error
        soft(e_flags, "pack flags");
    }
    fh_prof_begin(h, "allreduce");
    rc = fh_comm_allreduce_sum(h, pack, total);
    fh_prof_end(h);
    if (rc) return local_rc ? local_rc : rc;
    if (local_rc) {
        hipStreamSynchronize(h->stream);          // best effort: leave no work of ours queued behind the error
        return local_rc;
    }
    cplx* const qdst = rs ? rs->P : dQproj;
    if (h->real_projection) fh_launch_unpack_real(pack, qdst, nq, h->stream);
    else FH_CHECK(hipMemcpyAsync(qdst, pack, nq * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    off = nq;
    if (dzAq) { FH_CHECK(hipMemcpyAsync(dzAq, pack + off, nm * sizeof(double), hipMemcpyDeviceToDevice, h->stream)); off += nm; }
    if (dzSq) { FH_CHECK(hipMemcpyAsync(dzSq, pack + off, nm * sizeof(double), hipMemcpyDeviceToDevice, h->stream)); off += nm; }
    FH_CHECK(hipMemcpyAsync(flags.data(), pack + off, flags.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (node_status)
        for (int g = 0; g < ne; ++g)
            node_status[g] = flags[ne + g] > 0.0 ? (int)FEASTHIP_ERROR_LAPACK : (flags[g] > 0.0 ? (int)FEASTHIP_ERROR_NO_CONVERGENCE : 0);
    h->global_node_iters.assign(ne, 0);
    for (int g = 0; g < ne; ++g) h->global_node_iters[g] = (int)(flags[2 * (size_t)ne + g] + 0.5);
    if (flags[3 * (size_t)ne] > 0.0) { h->last_error = "contour_apply: the sweep failed on another rank"; return FEASTHIP_ERROR_INTERNAL; }
    return 0;
}

extern "C" int feasthip_set_column_block(feasthip_handle h, int64_t first, int64_t count) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (count < 0) { h->col_block_lo = 0; h->col_block_hi = -1; return 0; }
    if (first < 0) { h->last_error = "set_column_block: first must be >= 0"; return FEASTHIP_ERROR_M0; }
    h->col_block_lo = first; h->col_block_hi = first + count;
    return 0;
}

extern "C" int feasthip_contour_apply_dev(feasthip_handle h, int64_t m, const void* dQ, const double* ritz_lambda_host,
                                          void* dQproj, void* dzAq, void* dzSq, int* node_status, feasthip_stats* stats) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (!dQ || !dQproj) { h->last_error = "contour_apply: null Q/Qproj"; return FEASTHIP_ERROR_INTERNAL; }
    // the column mask is one-shot: it applies to this sweep only and never leaks into later calls
    // (shifted_solve / RCI jobs on the same handle), whatever the outcome of the sweep
    h->mask_live = 1;
    const int rc = fh_contour_apply_impl(h, m, (const cplx*)dQ, ritz_lambda_host, (cplx*)dQproj, (cplx*)dzAq, (cplx*)dzSq, node_status, stats);
    h->mask_live = 0;
    h->col_mask.clear();
    return rc;
}

// host-pointer wrapper helpers
static int fh_stage_in(feasthip_ctx* h, const char* name, const void* host, size_t bytes, void** dev) {
    int rc = fh_get_buf(h, name, bytes, dev);
    if (rc) return rc;
    FH_CHECK(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, h->stream));
    return 0;
}

extern "C" int feasthip_contour_apply(feasthip_handle h, int64_t m, const void* Q, const double* ritz_lambda,
                                      void* Qproj, void* zAq, void* zSq, int* node_status, feasthip_stats* stats) {
    int rc = fh_check_problem(h, m, 1);
    if (rc) return rc;
    if (!Q || !Qproj) { h->last_error = "contour_apply: null Q/Qproj"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const size_t nb = (size_t)fh_N(h) * m * sizeof(cplx), mb = (size_t)m * m * sizeof(cplx);
    void *dQ, *dP, *dA = nullptr, *dS = nullptr;
    if ((rc = fh_stage_in(h, "host_Q", Q, nb, &dQ))) return rc;
    if ((rc = fh_get_buf(h, "host_Qproj", nb, &dP))) return rc;
    if (zAq && (rc = fh_get_buf(h, "host_zAq", mb, &dA))) return rc;
    if (zSq && (rc = fh_get_buf(h, "host_zSq", mb, &dS))) return rc;
    h->mask_live = 1;
    rc = fh_contour_apply_impl(h, m, (const cplx*)dQ, ritz_lambda, (cplx*)dP, (cplx*)dA, (cplx*)dS, node_status, stats);
    h->mask_live = 0;
    h->col_mask.clear();
    if (rc) return rc;
    FH_CHECK(hipMemcpy(Qproj, dP, nb, hipMemcpyDeviceToHost));
    if (zAq) FH_CHECK(hipMemcpy(zAq, dA, mb, hipMemcpyDeviceToHost));
    if (zSq) FH_CHECK(hipMemcpy(zSq, dS, mb, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------
// small host-side complex Hermitian helpers (m <= 64) for the Cholesky-QR fast path
// ---------------------------------------------------------------------------------------
// pivoted Cholesky pivots of a Hermitian PSD matrix (column-major, leading dim ld): returns
// min/max pivot ratio.  In exact arithmetic these pivots are the squares of the diagonal of R
// in the column-pivoted QR of the panel, so the ratio bounds the rank test of
// _feast_qr_compress! (src/core/feast_aux.jl:117-124) from the safe side.
static double fh_pivoted_cholesky_ratio(std::vector<cplx> G, int m, int ld) {
    std::vector<int> perm(m);
    for (int i = 0; i < m; ++i) perm[i] = i;
    double dmax = 0.0, dmin = 0.0;
    auto at = [&](int i, int j) -> cplx& { return G[(size_t)j * ld + i]; };
    for (int k = 0; k < m; ++k) {
        int p = k;
        for (int j = k + 1; j < m; ++j) if (at(j, j).x > at(p, p).x) p = j;
        if (p != k) {     // symmetric swap of rows/cols k and p
            for (int j = 0; j < m; ++j) std::swap(at(k, j), at(p, j));
            for (int i = 0; i < m; ++i) std::swap(at(i, k), at(i, p));
        }
        double d = at(k, k).x;
        if (k == 0) dmax = d;
        if (!(d > 0.0) || !std::isfinite(d)) return 0.0;
        dmin = d;
        double r = std::sqrt(d);
        at(k, k) = cmake(r, 0);
        for (int i = k + 1; i < m; ++i) at(i, k) = cscale(at(i, k), 1.0 / r);
        for (int j = k + 1; j < m; ++j)
            for (int i = j; i < m; ++i) {
                cplx v = csub(at(i, j), cmul(at(i, k), cconj(at(j, k))));
                at(i, j) = v;
                at(j, i) = cconj(v);
            }
    }
    return dmax > 0.0 ? dmin / dmax : 0.0;
}

// Real twins of the two routines for a Gram matrix without imaginary parts (real projection of real-symmetric input:
// Q_proj is real): the same arithmetic on a quarter of the flops -- the host's share of the orthonormalisation was
// most of the 0.56 ms the step took per refinement loop on cfg 3.
static double fh_pivoted_cholesky_ratio_real(std::vector<double> G, int m) {
    double dmax = 0.0, dmin = 0.0;
    auto at = [&](int i, int j) -> double& { return G[(size_t)j * m + i]; };
    for (int k = 0; k < m; ++k) {
        int p = k;
        for (int j = k + 1; j < m; ++j) if (at(j, j) > at(p, p)) p = j;
        if (p != k) {
            for (int j = 0; j < m; ++j) std::swap(at(k, j), at(p, j));
            for (int i = 0; i < m; ++i) std::swap(at(i, k), at(i, p));
        }
        const double d = at(k, k);
        if (k == 0) dmax = d;
        if (!(d > 0.0) || !std::isfinite(d)) return 0.0;
        dmin = d;
        const double r = std::sqrt(d);
        for (int i = k + 1; i < m; ++i) at(i, k) /= r;
        for (int j = k + 1; j < m; ++j) {
            const double ajk = at(j, k);
            for (int i = j; i < m; ++i) { const double v = at(i, j) - at(i, k) * ajk; at(i, j) = v; at(j, i) = v; }
        }
    }
    return dmax > 0.0 ? dmin / dmax : 0.0;
}
static bool fh_chol_upper_inverse_real(const std::vector<double>& G, int m, int ld, std::vector<cplx>& Rinv) {
    std::vector<double> R((size_t)m * m, 0.0), X((size_t)m * m, 0.0);
    auto r = [&](int i, int j) -> double& { return R[(size_t)j * m + i]; };
    for (int j = 0; j < m; ++j)
        for (int i = 0; i <= j; ++i) {
            double sum = G[(size_t)j * m + i];
            for (int k = 0; k < i; ++k) sum -= r(k, i) * r(k, j);
            if (i == j) {
                if (!(sum > 0.0) || !std::isfinite(sum)) return false;
                r(i, i) = std::sqrt(sum);
            } else {
                r(i, j) = sum / r(i, i);
            }
        }
    Rinv.assign((size_t)ld * ld, cmake(0, 0));
    for (int j = 0; j < m; ++j) {
        X[(size_t)j * m + j] = 1.0 / r(j, j);
        for (int i = j - 1; i >= 0; --i) {
            double sum = 0.0;
            for (int k = i + 1; k <= j; ++k) sum += r(i, k) * X[(size_t)j * m + k];
            X[(size_t)j * m + i] = -sum / r(i, i);
        }
        for (int i = 0; i <= j; ++i) Rinv[(size_t)j * ld + i] = cmake(X[(size_t)j * m + i], 0);
    }
    return true;
}

// Rinv (ld x ld, column-major, zero padded) with G = R^H R, R upper triangular; false if not PD
static bool fh_chol_upper_inverse(const std::vector<cplx>& G, int m, int ld, std::vector<cplx>& Rinv) {
    std::vector<cplx> R((size_t)m * m, cmake(0, 0));
    auto r = [&](int i, int j) -> cplx& { return R[(size_t)j * m + i]; };
    for (int j = 0; j < m; ++j) {
        for (int i = 0; i <= j; ++i) {
            cplx sum = G[(size_t)j * ld + i];
            for (int k = 0; k < i; ++k) sum = csub(sum, cmul(cconj(r(k, i)), r(k, j)));
            if (i == j) {
                if (!(sum.x > 0.0) || !std::isfinite(sum.x)) return false;
                r(i, i) = cmake(std::sqrt(sum.x), 0);
            } else {
                r(i, j) = cscale(sum, 1.0 / r(i, i).x);
            }
        }
    }
    Rinv.assign((size_t)ld * ld, cmake(0, 0));
    for (int j = 0; j < m; ++j) {          // back substitution, column by column
        Rinv[(size_t)j * ld + j] = cmake(1.0 / r(j, j).x, 0);
        for (int i = j - 1; i >= 0; --i) {
            cplx sum = cmake(0, 0);
            for (int k = i + 1; k <= j; ++k) sum = cadd(sum, cmul(r(i, k), Rinv[(size_t)j * ld + k]));
            Rinv[(size_t)j * ld + i] = cscale(sum, -1.0 / r(i, i).x);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// orthonormalisation (a9)
// ---------------------------------------------------------------------------------------
// Rank-revealing orthonormalisation of the m (<= ld) columns of panel X.  On success *res is the
// panel (X or Out) whose first *rank columns hold the basis.  ref_scale > 0 / big_dim: X is a
// block of a wider matrix (see k_mgs_pick).
static int fh_ortho_panel(feasthip_ctx* h, int m, int ld, cplx* X, cplx* Out, double rank_tol, double ref_scale,
                          int big_dim, int* rank, cplx** res) {
    const int N = (int)fh_N(h);
    int rc;
    void* p;
    if ((rc = fh_get_buf(h, "or_work", (size_t)256 * ld * sizeof(cplx), &p))) return rc;
    cplx* work = (cplx*)p;
    if ((rc = fh_get_buf(h, "or_istate", (4 + FH_MAX_LD) * sizeof(int), &p))) return rc;
    int* istate = (int*)p;
    if ((rc = fh_get_buf(h, "or_dstate", (2 + FH_MAX_LD) * sizeof(double), &p))) return rc;
    double* dstate = (double*)p;
    if ((rc = fh_get_buf(h, "or_coef", FH_MAX_LD * sizeof(cplx), &p))) return rc;
    cplx* coef = (cplx*)p;
    // Fast path (Cholesky-QR twice) when the panel is far from rank deficient: the pivoted
    // Cholesky pivots of the Gram matrix are the squared R_kk of the pivoted QR, so a pivot
    // ratio above 1e-10 means every |R_kk|/|R_11| > 1e-5 >> rank_tol and the reference rule
    // keeps all m columns.  Otherwise fall through to the rank-revealing pivoted Gram-Schmidt.
    if (!getenv("FH_NO_CHOLQR")) {
        if ((rc = fh_get_buf(h, "gram_work", fh_gram_work_elems(ld) * sizeof(cplx), &p))) return rc;
        cplx* gw = (cplx*)p;
        if ((rc = fh_get_buf(h, "gram_G", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
        cplx* G = (cplx*)p;
        if ((rc = fh_get_buf(h, "or_Rinv", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
        cplx* dR = (cplx*)p;
        std::vector<cplx> Gh((size_t)ld * ld), Rinv;
        bool ok = true;
        cplx* src = X;
        cplx* dst = Out;
        // A second Cholesky-QR pass squares away the orthogonality error of the first, eps / ratio' (ratio' = pivot ratio
        // of the equilibrated Gram matrix ~ 1 / its condition number).  With ratio' > 1e-2 the first pass is already at
        // 1e-14 -- the FEAST panel in steady state (B-orthonormal Ritz vectors times filter values, equilibrated) has
        // ratio' ~ 0.4 -- and the Rayleigh-Ritz step that follows uses Q^H B Q anyway, so the pass is skipped.
        int npass = 2;
        static const bool always_two = getenv("FH_CHOLQR_TWO_PASS") != nullptr;
        for (int pass = 0; pass < npass && ok; ++pass) {
            fh_prof_begin(h, "gram");
            fh_launch_gram(src, src, N, ld, 0, gw, G, h->stream);
            fh_prof_end(h);
            FH_CHECK(hipMemcpyAsync(Gh.data(), G, Gh.size() * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
            FH_CHECK(hipStreamSynchronize(h->stream));
            std::vector<double> dcol;
            double dmin_eq = 1.0, dmax_eq = 1.0;
            if (pass == 0) {
                // Equilibrate: G = D G' D with D = diag(column norms).  Columns of very different
                // length (e.g. guard columns scaled by a small filter value) make G ill-conditioned
                // although the directions are fine; Cholesky of G' is as stable as for unit columns.
                // Full rank in the sense of the reference's pivoted-QR rule is accepted only with a
                // wide margin: |R_kk|/|R_11| >~ (d_min/d_max) sqrt(ratio') must exceed 1e3 rank_tol.
                dcol.resize(m);
                double dmin = 0.0, dmax = ref_scale;
                for (int j = 0; j < m; ++j) {
                    const double g = Gh[(size_t)j * ld + j].x;
                    dcol[j] = g > 0.0 && std::isfinite(g) ? std::sqrt(g) : 0.0;
                    dmin = j == 0 ? dcol[j] : std::min(dmin, dcol[j]);
                    dmax = std::max(dmax, dcol[j]);
                }
                if (!(dmin > 0.0)) { ok = false; break; }
                for (int j = 0; j < m; ++j)
                    for (int i = 0; i < m; ++i) {
                        cplx& g = Gh[(size_t)j * ld + i];
                        g = cscale(g, 1.0 / (dcol[i] * dcol[j]));
                    }
                dmin_eq = dmin; dmax_eq = dmax;
            }
            // a Gram matrix without imaginary parts (real Q_proj) takes the real routines
            bool is_real = true;
            for (int j = 0; j < m && is_real; ++j)
                for (int i = 0; i < m; ++i) if (Gh[(size_t)j * ld + i].y != 0.0) { is_real = false; break; }
            std::vector<double> Gr;
            if (is_real) {
                Gr.resize((size_t)m * m);
                for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) Gr[(size_t)j * m + i] = Gh[(size_t)j * ld + i].x;
            }
            if (pass == 0) {
                const double ratio = is_real ? fh_pivoted_cholesky_ratio_real(Gr, m) : fh_pivoted_cholesky_ratio(Gh, m, ld);
                if (!(ratio > 1e-10) || !((dmin_eq / dmax_eq) * std::sqrt(ratio) > 1e3 * rank_tol)) { ok = false; break; }
                if (ratio > 1e-2 && !always_two) npass = 1;
            }
            if (!(is_real ? fh_chol_upper_inverse_real(Gr, m, ld, Rinv) : fh_chol_upper_inverse(Gh, m, ld, Rinv))) { ok = false; break; }
            if (pass == 0)       // R = R' D  =>  R^-1 = D^-1 R'^-1: scale row i by 1/d_i
                for (int j = 0; j < m; ++j)
                    for (int i = 0; i < m; ++i) Rinv[(size_t)j * ld + i] = cscale(Rinv[(size_t)j * ld + i], 1.0 / dcol[i]);
            FH_CHECK(hipMemcpyAsync(dR, Rinv.data(), Rinv.size() * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
            fh_prof_begin(h, "ortho");
            fh_launch_small_matmul(src, dR, N, ld, dst, h->stream);
            fh_prof_end(h);
            FH_CHECK(hipStreamSynchronize(h->stream));     // Rinv (host vector) is reused
            std::swap(src, dst);
        }
        if (ok) {       // src after the swaps: X after two passes, Out after one
            *rank = m;
            *res = src;
            return 0;
        }
        // a failure happens before the pass writes its destination: X still holds the input
        // (pass 0 writes Out, pass 1 would have written X)
    }
    fh_mgs_args a;
    a.X = X; a.N = N; a.ld = ld; a.m = m; a.istate = istate; a.dstate = dstate; a.coef = coef; a.work = work;
    a.rank_tol = rank_tol; a.ref_scale = ref_scale; a.big_dim = big_dim;
    fh_prof_begin(h, "ortho");
    fh_mgs_run(a, h->stream);
    fh_prof_end(h);
    int hst[4 + FH_MAX_LD];
    FH_CHECK(hipMemcpyAsync(hst, istate, (4 + ld) * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    int r = hst[2] ? hst[1] : hst[0];
    if (r < 0) r = 0;
    if (r > m) r = m;
    *rank = r;
    fh_launch_gather_cols(X, istate + 4, r, N, ld, Out, h->stream);
    *res = Out;
    return 0;
}

// m > 64: block Gram-Schmidt over 64-column panels.  Panel j is projected twice against the kept
// columns (C = K^H X_j on the MFMA Gram kernel, X_j -= K C), then orthonormalised by fh_ortho_panel
// with the rank threshold tied to the largest column norm of the WHOLE matrix -- the R_11 of the
// reference's pivoted QR (src/core/feast_aux.jl:113-124).  Kept columns are packed to the front of
// Q.  Pivoting is per panel, not global: for a full-rank input the basis spans the same space.
static int fh_ortho_wide(feasthip_ctx* h, int m, cplx* dQ, double rank_tol, int* rank) {
    const int N = (int)fh_N(h), ld = FH_MAX_LD;
    const size_t panel = (size_t)N * ld;
    int rc;
    void* p;
    if ((rc = fh_get_buf(h, "or_X", panel * sizeof(cplx), &p))) return rc;
    cplx* X = (cplx*)p;
    if ((rc = fh_get_buf(h, "or_out", panel * sizeof(cplx), &p))) return rc;
    cplx* Out = (cplx*)p;
    if ((rc = fh_get_buf(h, "ow_K", panel * sizeof(cplx), &p))) return rc;
    cplx* K = (cplx*)p;
    if ((rc = fh_get_buf(h, "ow_T", panel * sizeof(cplx), &p))) return rc;
    cplx* T = (cplx*)p;
    if ((rc = fh_get_buf(h, "gram_work", fh_gram_work_elems(ld) * sizeof(cplx), &p))) return rc;
    cplx* gw = (cplx*)p;
    if ((rc = fh_get_buf(h, "ow_C", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
    cplx* C = (cplx*)p;
    if ((rc = fh_get_buf(h, "ow_part", (size_t)fh_vec_nblk(N, ld) * ld * sizeof(cplx), &p))) return rc;
    cplx* part = (cplx*)p;
    if ((rc = fh_get_buf(h, "ow_dots", (size_t)ld * sizeof(cplx), &p))) return rc;
    cplx* ddots = (cplx*)p;
    std::vector<cplx> ones(ld, cmake(1, 0));
    cplx* done;
    if ((rc = fh_upload_coefs(h, "ow_one", ones, &done))) return rc;
    const int npan = (m + ld - 1) / ld;
    // largest column norm of the whole matrix
    double ref = 0.0;
    std::vector<cplx> dots(ld);
    for (int j = 0; j < npan; ++j) {
        const int mj = std::min(ld, m - j * ld);
        fh_launch_to_panel(dQ + (size_t)j * ld * N, N, N, mj, X, ld, h->stream, fh_perm(h));
        fh_launch_dot_cols(X, X, N, ld, part, ddots, h->stream);
        FH_CHECK(hipMemcpyAsync(dots.data(), ddots, ld * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));
        for (int c = 0; c < mj; ++c) if (dots[c].x > ref * ref) ref = std::sqrt(dots[c].x);
    }
    int kept = 0;
    for (int j = 0; j < npan; ++j) {
        const int mj = std::min(ld, m - j * ld);
        fh_launch_to_panel(dQ + (size_t)j * ld * N, N, N, mj, X, ld, h->stream, fh_perm(h));
        for (int pass = 0; pass < 2; ++pass) {
            for (int k0 = 0; k0 < kept; k0 += ld) {
                const int kw = std::min(ld, kept - k0);
                fh_launch_to_panel(dQ + (size_t)k0 * N, N, N, kw, K, ld, h->stream, fh_perm(h));
                fh_prof_begin(h, "gram");
                fh_launch_gram(K, X, N, ld, 0, gw, C, h->stream);        // C = K^H X
                fh_prof_end(h);
                fh_prof_begin(h, "ortho");
                fh_launch_small_matmul(K, C, N, ld, T, h->stream);       // T = K C
                fh_launch_axpy_cols(X, T, done, N, ld, h->stream);       // X -= T
                fh_prof_end(h);
            }
        }
        int rj = 0;
        cplx* res = nullptr;
        if ((rc = fh_ortho_panel(h, mj, ld, X, Out, rank_tol, ref, m, &rj, &res))) return rc;
        if (rj > 0) fh_launch_from_panel(res, ld, N, rj, dQ + (size_t)kept * N, N, h->stream, fh_perm(h));
        kept += rj;
    }
    *rank = kept;
    FH_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int feasthip_orthonormalize_dev(feasthip_handle h, int64_t m64, void* dQ, double rank_tol, int* rank) {
    int rc = fh_check_problem(h, m64, 1);
    if (rc) return rc;
    if (!dQ || !rank) { h->last_error = "orthonormalize: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    if (m64 > FH_MAX_LD) {
        rc = fh_ortho_wide(h, (int)m64, (cplx*)dQ, rank_tol, rank);
        fh_prof_collect(h);
        return rc;
    }
    const int m = (int)m64, ld = fh_pick_ld(m), N = (int)fh_N(h);
    const size_t panel = (size_t)N * ld;
    void* p;
    if ((rc = fh_get_buf(h, "or_X", panel * sizeof(cplx), &p))) return rc;
    cplx* X = (cplx*)p;
    if ((rc = fh_get_buf(h, "or_out", panel * sizeof(cplx), &p))) return rc;
    cplx* Out = (cplx*)p;
    fh_launch_to_panel((const cplx*)dQ, N, N, m, X, ld, h->stream, fh_perm(h));
    cplx* res = nullptr;
    if ((rc = fh_ortho_panel(h, m, ld, X, Out, rank_tol, 0.0, m, rank, &res))) return rc;
    fh_launch_from_panel(res, ld, N, m, (cplx*)dQ, N, h->stream, fh_perm(h));
    FH_CHECK(hipStreamSynchronize(h->stream));
    fh_prof_collect(h);
    FH_CHECK(hipGetLastError());       // launch-configuration errors do not surface through the stream sync
    return 0;
}

extern "C" int feasthip_orthonormalize(feasthip_handle h, int64_t m, void* Q, double rank_tol, int* rank) {
    int rc = fh_check_problem(h, m, 1);
    if (rc) return rc;
    if (!Q || !rank) { h->last_error = "orthonormalize: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const size_t nb = (size_t)fh_N(h) * m * sizeof(cplx);
    void* dQ;
    if ((rc = fh_stage_in(h, "host_Q", Q, nb, &dQ))) return rc;
    rc = feasthip_orthonormalize_dev(h, m, dQ, rank_tol, rank);
    if (rc) return rc;
    FH_CHECK(hipMemcpy(Q, dQ, nb, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------
// Rayleigh-Ritz projection (a10)
// ---------------------------------------------------------------------------------------
static void fh_hermitize(std::vector<cplx>& G, int r) {
    // _feast_hermitian_part!  src/core/feast_aux.jl:84-92
    std::vector<cplx> out((size_t)r * r);
    for (int j = 0; j < r; ++j)
        for (int i = 0; i < r; ++i) {
            cplx a = G[(size_t)j * r + i], b = cconj(G[(size_t)i * r + j]);
            out[(size_t)j * r + i] = cmake(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
        }
    G.swap(out);
}

extern "C" int feasthip_project_dev(feasthip_handle h, int64_t r64, const void* dQ, int bilinear, int hermitize,
                                    void* Aq_host, void* Bq_host) {
    if (r64 > FH_MAX_LD) {
        // r > 64: 64-column panels; block (i, j) of Q^H A Q is Q_i^H (A Q_j) on the MFMA Gram kernel
        int rc0 = fh_check_problem(h, r64, 1);
        if (rc0) return rc0;
        if (!dQ || !Aq_host) { h->last_error = "project: null argument"; return FEASTHIP_ERROR_INTERNAL; }
        FH_CHECK(hipSetDevice(h->device));
        const int r = (int)r64, ld = FH_MAX_LD, N = (int)fh_N(h);
        const size_t panel = (size_t)N * ld;
        void* p;
        int rc;
        if ((rc = fh_get_buf(h, "pj_Q", panel * sizeof(cplx), &p))) return rc;
        cplx* Qi = (cplx*)p;
        if ((rc = fh_get_buf(h, "pj_Qj", panel * sizeof(cplx), &p))) return rc;
        cplx* Qj = (cplx*)p;
        if ((rc = fh_get_buf(h, "pj_W", panel * sizeof(cplx), &p))) return rc;
        cplx* W = (cplx*)p;
        if ((rc = fh_get_buf(h, "gram_work", fh_gram_work_elems(ld) * sizeof(cplx), &p))) return rc;
        cplx* gw = (cplx*)p;
        if ((rc = fh_get_buf(h, "gram_G", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
        cplx* G = (cplx*)p;
        std::vector<cplx> one(ld, cmake(1, 0)), zero(ld, cmake(0, 0));
        cplx *d1, *d0;
        if ((rc = fh_upload_coefs(h, "pj_one", one, &d1))) return rc;
        if ((rc = fh_upload_coefs(h, "pj_zero", zero, &d0))) return rc;
        const int npan = (r + ld - 1) / ld;
        std::vector<cplx> Gh((size_t)ld * ld);
        for (int which = 0; which < 2; ++which) {
            cplx* out_host = (cplx*)(which == 0 ? Aq_host : Bq_host);
            if (!out_host) continue;
            std::vector<cplx> res((size_t)r * r, cmake(0, 0));
            if (which == 1 && fh_b_identity(h) && hermitize && !bilinear) {
                for (int i = 0; i < r; ++i) res[(size_t)i * r + i] = cmake(1, 0);
            } else {
                for (int j = 0; j < npan; ++j) {
                    const int mj = std::min(ld, r - j * ld);
                    fh_launch_to_panel((const cplx*)dQ + (size_t)j * ld * N, N, N, mj, Qj, ld, h->stream, fh_perm(h));
                    fh_op_call oc;
                    oc.m = mj;
                    oc.X = Qj; oc.x_stride = 0; oc.Y = W; oc.y_stride = 0;
                    oc.coefA = which == 0 ? d1 : d0; oc.coefB = which == 0 ? d0 : d1;
                    oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
                    oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
                    fh_apply_operator(h, ld, oc);
                    for (int i = 0; i < npan; ++i) {
                        const int mi = std::min(ld, r - i * ld);
                        fh_launch_to_panel((const cplx*)dQ + (size_t)i * ld * N, N, N, mi, Qi, ld, h->stream, fh_perm(h));
                        fh_prof_begin(h, "gram");
                        fh_launch_gram(Qi, W, N, ld, bilinear, gw, G, h->stream);
                        fh_prof_end(h);
                        FH_CHECK(hipMemcpyAsync(Gh.data(), G, Gh.size() * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
                        FH_CHECK(hipStreamSynchronize(h->stream));
                        for (int c2 = 0; c2 < mj; ++c2)
                            for (int c1 = 0; c1 < mi; ++c1)
                                res[(size_t)(j * ld + c2) * r + i * ld + c1] = Gh[(size_t)c2 * ld + c1];
                    }
                }
                if (hermitize && !bilinear) fh_hermitize(res, r);
            }
            memcpy(out_host, res.data(), res.size() * sizeof(cplx));
        }
        fh_prof_collect(h);
        return 0;
    }

    int rc = fh_check_problem(h, r64);
    if (rc) return rc;
    if (!dQ || !Aq_host) { h->last_error = "project: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const int r = (int)r64, ld = fh_pick_ld(r), N = (int)fh_N(h);
    const size_t panel = (size_t)N * ld;
    void* p;
    if ((rc = fh_get_buf(h, "pj_Q", panel * sizeof(cplx), &p))) return rc;
    cplx* Qp = (cplx*)p;
    if ((rc = fh_get_buf(h, "pj_W", panel * sizeof(cplx), &p))) return rc;
    cplx* W = (cplx*)p;
    if ((rc = fh_get_buf(h, "gram_work", fh_gram_work_elems(ld) * sizeof(cplx), &p))) return rc;
    cplx* gw = (cplx*)p;
    if ((rc = fh_get_buf(h, "gram_G", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
    cplx* G = (cplx*)p;
    fh_launch_to_panel((const cplx*)dQ, N, N, r, Qp, ld, h->stream, fh_perm(h));
    std::vector<cplx> one(ld, cmake(1, 0)), zero(ld, cmake(0, 0));
    cplx *d1, *d0;
    if ((rc = fh_upload_coefs(h, "pj_one", one, &d1))) return rc;
    if ((rc = fh_upload_coefs(h, "pj_zero", zero, &d0))) return rc;
    // both Gram matrices are queued before the ONE synchronisation that brings them to the host
    if ((rc = fh_get_buf(h, "gram_G2", 2 * (size_t)ld * ld * sizeof(cplx), &p))) return rc;
    G = (cplx*)p;
    std::vector<cplx> Gh(2 * (size_t)ld * ld);
    bool queued[2] = {false, false};
    for (int which = 0; which < 2; ++which) {
        cplx* out_host = (cplx*)(which == 0 ? Aq_host : Bq_host);
        if (!out_host) continue;
        if (which == 1 && fh_b_identity(h) && hermitize && !bilinear) continue;       // Aq_rank = I exactly, below
        // (B = I without orthonormal Q, variant C: the operator kernel yields W = Q, so G = Q^H Q)
        fh_op_call oc;
        oc.m = r;
        oc.X = Qp; oc.x_stride = 0; oc.Y = W; oc.y_stride = 0;
        oc.coefA = which == 0 ? d1 : d0; oc.coefB = which == 0 ? d0 : d1;
        oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
        oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
        fh_apply_operator(h, ld, oc);
        fh_prof_begin(h, "gram");
        fh_launch_gram(Qp, W, N, ld, bilinear, gw, G + (size_t)which * ld * ld, h->stream);
        fh_prof_end(h);
        FH_CHECK(hipMemcpyAsync(Gh.data() + (size_t)which * ld * ld, G + (size_t)which * ld * ld, (size_t)ld * ld * sizeof(cplx),
                                hipMemcpyDeviceToHost, h->stream));
        queued[which] = true;
    }
    FH_CHECK(hipStreamSynchronize(h->stream));
    for (int which = 0; which < 2; ++which) {
        cplx* out_host = (cplx*)(which == 0 ? Aq_host : Bq_host);
        if (!out_host) continue;
        std::vector<cplx> res((size_t)r * r);
        if (!queued[which]) {
            // variant A: Q is orthonormal, Aq_rank = I exactly (src/dense/feast_dense.jl:255-259)
            for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) res[(size_t)j * r + i] = cmake(i == j ? 1.0 : 0.0, 0.0);
        } else {
            const cplx* Gw = Gh.data() + (size_t)which * ld * ld;
            for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) res[(size_t)j * r + i] = Gw[(size_t)j * ld + i];
            if (hermitize && !bilinear) fh_hermitize(res, r);
        }
        memcpy(out_host, res.data(), res.size() * sizeof(cplx));
    }
    fh_prof_collect(h);
    FH_CHECK(hipGetLastError());       // launch-configuration errors do not surface through the stream sync
    return 0;
}

extern "C" int feasthip_project(feasthip_handle h, int64_t r, const void* Q, int bilinear, int hermitize, void* Aq, void* Bq) {
    int rc = fh_check_problem(h, r, 1);
    if (rc) return rc;
    if (!Q) { h->last_error = "project: null Q"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    void* dQ;
    if ((rc = fh_stage_in(h, "host_Q", Q, (size_t)fh_N(h) * r * sizeof(cplx), &dQ))) return rc;
    return feasthip_project_dev(h, r, dQ, bilinear, hermitize, Aq, Bq);
}

// ---------------------------------------------------------------------------------------
// Ritz back-transform + residual (a12, a13)
// ---------------------------------------------------------------------------------------
extern "C" int feasthip_ritz_residual_dev(feasthip_handle h, int64_t r64, const void* dQ, const void* V_host,
                                          const double* lambda_host, int64_t M, int normalize, int use_B, void* dX,
                                          double* res_host) {
    if (r64 > FH_MAX_LD) {
        // r > 64: X_j = sum_i Q_i V[i-block, j-block] per 64-column output panel, then the panel
        // goes through the same normalise / residual steps as the narrow path
        int rc0 = fh_check_problem(h, r64, 1);
        if (rc0) return rc0;
        if (!dQ || !V_host || !lambda_host || !dX) { h->last_error = "ritz_residual: null argument"; return FEASTHIP_ERROR_INTERNAL; }
        if (M < 0 || M > r64) { h->last_error = "ritz_residual: M out of range"; return FEASTHIP_ERROR_M0; }
        if (dQ == dX) { h->last_error = "ritz_residual: X must not alias Q for r > 64"; return FEASTHIP_ERROR_INTERNAL; }
        FH_CHECK(hipSetDevice(h->device));
        const int r = (int)r64, ld = FH_MAX_LD, N = (int)fh_N(h);
        const size_t panel = (size_t)N * ld;
        void* p;
        int rc;
        if ((rc = fh_get_buf(h, "rz_Q", panel * sizeof(cplx), &p))) return rc;
        cplx* Qp = (cplx*)p;
        if ((rc = fh_get_buf(h, "rz_X", panel * sizeof(cplx), &p))) return rc;
        cplx* Xp = (cplx*)p;
        if ((rc = fh_get_buf(h, "rz_R", panel * sizeof(cplx), &p))) return rc;
        cplx* Rp = (cplx*)p;
        if ((rc = fh_get_buf(h, "rz_T", panel * sizeof(cplx), &p))) return rc;
        cplx* Tp = (cplx*)p;
        if ((rc = fh_get_buf(h, "rz_V", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
        cplx* dV = (cplx*)p;
        const int nblk_op = fh_op_nblk(h, ld), nblk_vec = fh_vec_nblk(N, ld);
        if ((rc = fh_get_buf(h, "rz_part", (size_t)std::max(nblk_op, nblk_vec) * ld * sizeof(cplx), &p))) return rc;
        cplx* part = (cplx*)p;
        if ((rc = fh_get_buf(h, "rz_dots", (size_t)ld * sizeof(cplx), &p))) return rc;
        cplx* ddots = (cplx*)p;
        std::vector<cplx> mones(ld, cmake(-1, 0));
        cplx* dmone;
        if ((rc = fh_upload_coefs(h, "rz_mone", mones, &dmone))) return rc;
        const cplx* Vh = (const cplx*)V_host;
        const int npan = (r + ld - 1) / ld;
        std::vector<cplx> Vp((size_t)ld * ld), dots(ld);
        for (int j = 0; j < npan; ++j) {
            const int mj = std::min(ld, r - j * ld);
            for (int i = 0; i < npan; ++i) {
                const int mi = std::min(ld, r - i * ld);
                std::fill(Vp.begin(), Vp.end(), cmake(0, 0));
                for (int c2 = 0; c2 < mj; ++c2)
                    for (int c1 = 0; c1 < mi; ++c1) Vp[(size_t)c2 * ld + c1] = Vh[(size_t)(j * ld + c2) * r + i * ld + c1];
                FH_CHECK(hipMemcpyAsync(dV, Vp.data(), Vp.size() * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
                fh_launch_to_panel((const cplx*)dQ + (size_t)i * ld * N, N, N, mi, Qp, ld, h->stream, fh_perm(h));
                fh_prof_begin(h, "ritz");
                if (i == 0) {
                    fh_launch_small_matmul(Qp, dV, N, ld, Xp, h->stream);
                } else {
                    fh_launch_small_matmul(Qp, dV, N, ld, Tp, h->stream);
                    fh_launch_axpy_cols(Xp, Tp, dmone, N, ld, h->stream);      // X += T
                }
                fh_prof_end(h);
                FH_CHECK(hipStreamSynchronize(h->stream));                      // Vp (host) is reused
            }
            const int Mj = std::max(0, std::min(mj, (int)M - j * ld));          // columns of this panel below M
            if (normalize && Mj > 0) {
                fh_launch_dot_cols(Xp, Xp, N, ld, part, ddots, h->stream);
                FH_CHECK(hipMemcpyAsync(dots.data(), ddots, ld * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
                FH_CHECK(hipStreamSynchronize(h->stream));
                std::vector<cplx> sc(ld, cmake(1, 0));
                for (int c = 0; c < Mj; ++c) {
                    double n = std::sqrt(dots[c].x);
                    if (n > 0) sc[c] = cmake(1.0 / n, 0);
                }
                cplx* dsc;
                if ((rc = fh_upload_coefs(h, "rz_scale", sc, &dsc))) return rc;
                fh_launch_scale_cols(Xp, dsc, N, ld, h->stream);
            }
            fh_launch_from_panel(Xp, ld, N, mj, (cplx*)dX + (size_t)j * ld * N, N, h->stream, fh_perm(h));
            if (Mj > 0 && res_host) {
                const double* lam = lambda_host + 2 * (size_t)j * ld;
                std::vector<cplx> ca(ld, cmake(1, 0)), cb(ld, cmake(0, 0));
                const bool lam_in_op = use_B || fh_b_identity(h);
                for (int c = 0; c < mj; ++c) cb[c] = lam_in_op ? cmake(-lam[2 * c], -lam[2 * c + 1]) : cmake(0, 0);
                cplx *dca, *dcb;
                if ((rc = fh_upload_coefs(h, "rz_coefA", ca, &dca))) return rc;
                if ((rc = fh_upload_coefs(h, "rz_coefB", cb, &dcb))) return rc;
                fh_op_call oc;
                oc.m = mj;
                oc.X = Xp; oc.x_stride = 0; oc.Y = Rp; oc.y_stride = 0; oc.coefA = dca; oc.coefB = dcb;
                oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
                oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
                fh_apply_operator(h, ld, oc);
                if (!use_B && !fh_b_identity(h)) {
                    std::vector<cplx> lv(ld, cmake(0, 0));
                    for (int c = 0; c < mj; ++c) lv[c] = cmake(lam[2 * c], lam[2 * c + 1]);
                    cplx* dl;
                    if ((rc = fh_upload_coefs(h, "rz_lam", lv, &dl))) return rc;
                    fh_launch_axpy_cols(Rp, Xp, dl, N, ld, h->stream);
                }
                fh_launch_dot_cols(Rp, Rp, N, ld, part, ddots, h->stream);
                FH_CHECK(hipMemcpyAsync(dots.data(), ddots, ld * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
                FH_CHECK(hipStreamSynchronize(h->stream));
                for (int c = 0; c < Mj; ++c) {
                    double la = std::hypot(lam[2 * c], lam[2 * c + 1]);
                    res_host[j * ld + c] = std::sqrt(dots[c].x) / std::max(la, 1.0);
                }
            }
        }
        FH_CHECK(hipStreamSynchronize(h->stream));
        fh_prof_collect(h);
        return 0;
    }

    int rc = fh_check_problem(h, r64);
    if (rc) return rc;
    if (!dQ || !V_host || !lambda_host || !dX) { h->last_error = "ritz_residual: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    if (M < 0 || M > r64) { h->last_error = "ritz_residual: M out of range"; return FEASTHIP_ERROR_M0; }
    FH_CHECK(hipSetDevice(h->device));
    const int r = (int)r64, ld = fh_pick_ld(r), N = (int)fh_N(h);
    const size_t panel = (size_t)N * ld;
    void* p;
    if ((rc = fh_get_buf(h, "rz_Q", panel * sizeof(cplx), &p))) return rc;
    cplx* Qp = (cplx*)p;
    if ((rc = fh_get_buf(h, "rz_X", panel * sizeof(cplx), &p))) return rc;
    cplx* Xp = (cplx*)p;
    if ((rc = fh_get_buf(h, "rz_R", panel * sizeof(cplx), &p))) return rc;
    cplx* Rp = (cplx*)p;
    if ((rc = fh_get_buf(h, "rz_V", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
    cplx* dV = (cplx*)p;
    const int nblk_op = fh_op_nblk(h, ld), nblk_vec = fh_vec_nblk(N, ld);
    if ((rc = fh_get_buf(h, "rz_part", (size_t)std::max(nblk_op, nblk_vec) * ld * sizeof(cplx), &p))) return rc;
    cplx* part = (cplx*)p;
    if ((rc = fh_get_buf(h, "rz_dots", (size_t)ld * sizeof(cplx), &p))) return rc;
    cplx* ddots = (cplx*)p;
    // V padded to ld x ld
    std::vector<cplx> Vp((size_t)ld * ld, cmake(0, 0));
    const cplx* Vh = (const cplx*)V_host;
    for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) Vp[(size_t)j * ld + i] = Vh[(size_t)j * r + i];
    FH_CHECK(hipMemcpyAsync(dV, Vp.data(), Vp.size() * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
    fh_launch_to_panel((const cplx*)dQ, N, N, r, Qp, ld, h->stream, fh_perm(h));
    fh_prof_begin(h, "ritz");
    fh_launch_small_matmul(Qp, dV, N, ld, Xp, h->stream);
    fh_prof_end(h);
    std::vector<cplx> dots(ld);
    if (normalize && M > 0) {
        // normalise the first M columns (src/dense/feast_dense.jl:301-305): norms and scaling stay on the device
        fh_launch_dot_cols(Xp, Xp, N, ld, part, ddots, h->stream);
        fh_launch_normalize_cols(Xp, ddots, N, ld, (int)M, h->stream);
    }
    fh_launch_from_panel(Xp, ld, N, r, (cplx*)dX, N, h->stream, fh_perm(h));
    if (M > 0 && res_host) {
        // R = A X - B X diag(lambda); res_j = ||R_j|| / max(|lambda_j|, 1)
        std::vector<cplx> ca(ld, cmake(1, 0)), cb(ld, cmake(0, 0));
        const bool lam_in_op = use_B || fh_b_identity(h);
        for (int c = 0; c < r; ++c) cb[c] = lam_in_op ? cmake(-lambda_host[2 * c], -lambda_host[2 * c + 1]) : cmake(0, 0);
        cplx *dca, *dcb;
        if ((rc = fh_upload_coefs(h, "rz_coefA", ca, &dca))) return rc;
        if ((rc = fh_upload_coefs(h, "rz_coefB", cb, &dcb))) return rc;
        fh_op_call oc;
        oc.m = r;
        oc.X = Xp; oc.x_stride = 0; oc.Y = Rp; oc.y_stride = 0; oc.coefA = dca; oc.coefB = dcb;
        oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
        oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
        fh_apply_operator(h, ld, oc);
        if (!use_B && !fh_b_identity(h)) {
            // RCI-style residual without B: R = A X - X diag(lambda)  (src/kernel/feast_kernel.jl:899-906)
            std::vector<cplx> lam(ld, cmake(0, 0));
            for (int c = 0; c < r; ++c) lam[c] = cmake(lambda_host[2 * c], lambda_host[2 * c + 1]);
            cplx* dl;
            if ((rc = fh_upload_coefs(h, "rz_lam", lam, &dl))) return rc;
            fh_launch_axpy_cols(Rp, Xp, dl, N, ld, h->stream);   // R -= X diag(lam)
        }
        fh_launch_dot_cols(Rp, Rp, N, ld, part, ddots, h->stream);
        FH_CHECK(hipMemcpyAsync(dots.data(), ddots, ld * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));
        for (int c = 0; c < (int)M; ++c) {
            double la = std::hypot(lambda_host[2 * c], lambda_host[2 * c + 1]);
            res_host[c] = std::sqrt(dots[c].x) / std::max(la, 1.0);
        }
    }
    FH_CHECK(hipStreamSynchronize(h->stream));
    fh_prof_collect(h);
    FH_CHECK(hipGetLastError());       // launch-configuration errors do not surface through the stream sync
    return 0;
}

extern "C" int feasthip_ritz_residual(feasthip_handle h, int64_t r, const void* Q, const void* V, const double* lambda,
                                      int64_t M, int normalize, int use_B, void* X, double* res) {
    int rc = fh_check_problem(h, r, 1);
    if (rc) return rc;
    if (!Q || !X) { h->last_error = "ritz_residual: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const size_t nb = (size_t)fh_N(h) * r * sizeof(cplx);
    void *dQ, *dX;
    if ((rc = fh_stage_in(h, "host_Q", Q, nb, &dQ))) return rc;
    if ((rc = fh_get_buf(h, "host_X", nb, &dX))) return rc;
    rc = feasthip_ritz_residual_dev(h, r, dQ, V, lambda, M, normalize, use_B, dX, res);
    if (rc) return rc;
    FH_CHECK(hipMemcpy(X, dX, nb, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------
// Resident refinement loop (rows a7, a9-a13 of one FEAST loop without leaving the kernels' panel layout)
//
// The per-primitive entry points above speak column-major at the C ABI, so one loop of variant A crossed the boundary seven
// times (contour_apply in/out, orthonormalize in/out, project in, ritz_residual in/out: k_to_panel / k_from_panel each) and
// synchronised the stream about twenty times (every small host -> device upload).  Here the panels stay where the kernels
// left them:
//   feasthip_contour_apply_resident   Q (imported once, or the Ritz vectors of the previous loop)  ->  Q_proj   [rs_P]
//   feasthip_rr_reduce_resident       rank + the reduced pencil  (Q_o^H A Q_o, Q_o^H B Q_o)  of the orthonormal basis Q_o of Q_proj
//   feasthip_rr_ritz_resident         X = Q_o V, normalise, residuals; X becomes the next loop's Q   [rs_X, rs_R]
//   feasthip_resident_export          column-major copy of X (the converged Ritz vectors, once per solve)
// The orthonormal basis is never formed when Q_proj is well conditioned (the steady state of FEAST): the Rayleigh-Ritz
// pairs of a subspace do not depend on the basis, so the reduced pencil is taken on Q_proj with unit columns -- three Gram
// products (Q^H Q for the test, Q^H A Q, Q^H B Q) queued behind ONE synchronisation, an O(m^2) scaling on the host -- and
// handed to the host's generalized eigensolver, whose Cholesky factorisation of the B-part does what the Cholesky-QR did;
// the Ritz vectors are Q_proj (D^-1 V): one tall product instead of two.  The acceptance test is fh_ortho_panel's one-pass
// condition (equilibrated pivoted-Cholesky ratio > 1e-2, the reference's rank rule with its margin); anything else -- rank
// deficiency, a ratio that needs the second Cholesky-QR pass -- takes fh_ortho_panel itself on the resident panel, so the
// rank decisions are the same as the per-primitive path's.
// The eigen-residual panel A X - B X diag(lambda) the Ritz step forms for its norms is the next sweep's shared start
// residual (fh_contour_apply_panel: shared_src), which saves that sweep's first operator product.
// ---------------------------------------------------------------------------------------
static int fh_rs_panels(feasthip_ctx* h, cplx** P, cplx** X, cplx** R) {
    const size_t bytes = (size_t)fh_N(h) * FH_MAX_LD * sizeof(cplx);
    void* p;
    int rc;
    if ((rc = fh_get_buf(h, "rs_P", bytes, &p))) return rc;
    *P = (cplx*)p;
    if ((rc = fh_get_buf(h, "rs_X", bytes, &p))) return rc;
    *X = (cplx*)p;
    if ((rc = fh_get_buf(h, "rs_R", bytes, &p))) return rc;
    *R = (cplx*)p;
    return 0;
}

extern "C" int feasthip_contour_apply_resident(feasthip_handle h, int64_t m64, const void* dQ, const double* ritz_lambda_host,
                                               int* node_status, feasthip_stats* stats) {
    int rc = fh_check_problem(h, m64);
    if (rc) return rc;
    if (h->zne.empty()) { h->last_error = "no contour set"; return FEASTHIP_ERROR_FPM; }
    FH_CHECK(hipSetDevice(h->device));
    const int m = (int)m64, ld = fh_pick_ld(m), N = (int)fh_N(h);
    cplx *P, *X, *R;
    if ((rc = fh_rs_panels(h, &P, &X, &R))) return rc;
    fh_resident_sweep rs;
    rs.P = P; rs.ld = ld; rs.eigres = nullptr;
    if (dQ) {
        void* p;
        if ((rc = fh_get_buf(h, "rs_Q0", (size_t)N * FH_MAX_LD * sizeof(cplx), &p))) return rc;
        fh_launch_to_panel((const cplx*)dQ, N, N, m, (cplx*)p, ld, h->stream, fh_perm(h));
        rs.Q = (const cplx*)p;
    } else {
        if (h->rs_X != X || h->rs_X_m != m || h->rs_X_ld != ld) {
            h->last_error = "contour_apply_resident: no resident Ritz vectors of this width (pass Q, or run rr_ritz_resident first)";
            return FEASTHIP_ERROR_M0;
        }
        rs.Q = X;
        if (ritz_lambda_host && h->rs_R == R && (int)h->rs_R_lambda.size() >= m) {
            bool same = true;
            for (int c = 0; c < m && same; ++c) same = h->rs_R_lambda[c].y == 0.0 && h->rs_R_lambda[c].x == ritz_lambda_host[c];
            if (same) rs.eigres = R;
        }
    }
    h->rs_P = nullptr; h->rs_basis = nullptr; h->rs_T.clear(); h->rs_rank = 0;
    h->mask_live = 1;
    rc = fh_contour_apply_impl(h, m64, nullptr, ritz_lambda_host, nullptr, nullptr, nullptr, node_status, stats, &rs);
    h->mask_live = 0;
    h->col_mask.clear();
    if (rc) return rc;
    h->rs_P = P; h->rs_m = m; h->rs_ld = ld;
    return 0;
}

extern "C" int feasthip_rr_reduce_resident(feasthip_handle h, int64_t m64, double rank_tol, int hermitize, int* rank,
                                           void* Aq_host, void* Bq_host) {
    int rc = fh_check_problem(h, m64);
    if (rc) return rc;
    if (!rank || !Aq_host || !Bq_host) { h->last_error = "rr_reduce_resident: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const int m = (int)m64, N = (int)fh_N(h);
    cplx *Pb, *Xb, *Rb;
    if ((rc = fh_rs_panels(h, &Pb, &Xb, &Rb))) return rc;
    if (h->rs_P != Pb || h->rs_m != m) { h->last_error = "rr_reduce_resident: no resident Q_proj of this width"; return FEASTHIP_ERROR_M0; }
    const int ld = h->rs_ld;
    cplx* P = h->rs_P;
    const size_t panel = (size_t)N * ld, g2 = (size_t)ld * ld;
    void* p;
    if ((rc = fh_get_buf(h, "pj_W", panel * sizeof(cplx), &p))) return rc;
    cplx* W = (cplx*)p;
    if ((rc = fh_get_buf(h, "gram_work", fh_gram_work_elems(ld) * sizeof(cplx), &p))) return rc;
    cplx* gw = (cplx*)p;
    if ((rc = fh_get_buf(h, "gram_G3", 3 * g2 * sizeof(cplx), &p))) return rc;
    cplx* G = (cplx*)p;
    std::vector<cplx> one(ld, cmake(1, 0)), zero(ld, cmake(0, 0));
    cplx *d1, *d0;
    if ((rc = fh_upload_coefs(h, "pj_one", one, &d1))) return rc;
    if ((rc = fh_upload_coefs(h, "pj_zero", zero, &d0))) return rc;
    const bool b_id = fh_b_identity(h);
    // Gram products of `basis`: [0] basis^H basis (want_g0), [1] basis^H A basis, [2] basis^H B basis (B != I); one sync
    const cplx* Gh = nullptr;
    std::vector<char> gh_fallback;
    auto grams = [&](const cplx* basis, bool want_g0) -> int {
        if (want_g0) { fh_prof_begin(h, "gram"); fh_launch_gram(basis, basis, N, ld, 0, gw, G, h->stream); fh_prof_end(h); }
        for (int which = 0; which < 2; ++which) {
            if (which == 1 && b_id) break;
            fh_op_call oc;
            oc.m = m;
            oc.X = basis; oc.x_stride = 0; oc.Y = W; oc.y_stride = 0;
            oc.coefA = which == 0 ? d1 : d0; oc.coefB = which == 0 ? d0 : d1;
            oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
            oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
            fh_apply_operator(h, ld, oc);
            fh_prof_begin(h, "gram");
            fh_launch_gram(basis, W, N, ld, 0, gw, G + (size_t)(1 + which) * g2, h->stream);
            fh_prof_end(h);
        }
        const void* slot = nullptr;
        int drc = fh_download_small(h, G, 3 * g2 * sizeof(cplx), &slot, gh_fallback);
        if (drc) return drc;
        FH_CHECK(hipStreamSynchronize(h->stream));
        Gh = (const cplx*)slot;
        return 0;
    };
    // out (r x r, column-major) = Gsrc scaled by 1/(d_i d_j) (d == null: as is), Hermitian part when asked
    auto emit = [&](const cplx* Gsrc, const double* d, int r, void* out_host) {
        cplx* out = (cplx*)out_host;
        for (int j = 0; j < r; ++j)
            for (int i = 0; i < r; ++i) {
                cplx g = Gsrc[(size_t)j * ld + i];
                if (d) g = cscale(g, 1.0 / (d[i] * d[j]));
                out[(size_t)j * r + i] = g;
            }
        if (hermitize)
            for (int j = 0; j < r; ++j)
                for (int i = 0; i <= j; ++i) {
                    const cplx a = out[(size_t)j * r + i], b = cconj(out[(size_t)i * r + j]);
                    const cplx hm = cmake(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
                    out[(size_t)j * r + i] = hm; out[(size_t)i * r + j] = cconj(hm);
                }
    };
    auto identity = [&](int r, void* out_host) {
        cplx* out = (cplx*)out_host;
        for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) out[(size_t)j * r + i] = cmake(i == j ? 1.0 : 0.0, 0.0);
    };
    if ((rc = grams(P, true))) return rc;
    // ---- implicit basis: the acceptance test of fh_ortho_panel, on the Gram matrix we already have ----
    bool fast = !getenv("FH_NO_CHOLQR");
    std::vector<double> dcol(m, 1.0);
    if (fast) {
        std::vector<cplx> G0(Gh, Gh + g2);
        double dmin = 0.0, dmax = 0.0;
        for (int j = 0; j < m; ++j) {
            const double g = G0[(size_t)j * ld + j].x;
            dcol[j] = g > 0.0 && std::isfinite(g) ? std::sqrt(g) : 0.0;
            dmin = j == 0 ? dcol[j] : std::min(dmin, dcol[j]);
            dmax = std::max(dmax, dcol[j]);
        }
        fast = dmin > 0.0;
        if (fast) {
            bool is_real = true;
            for (int j = 0; j < m; ++j)
                for (int i = 0; i < m; ++i) {
                    cplx& g = G0[(size_t)j * ld + i];
                    g = cscale(g, 1.0 / (dcol[i] * dcol[j]));
                    if (g.y != 0.0) is_real = false;
                }
            double ratio;
            if (is_real) {
                std::vector<double> Gr((size_t)m * m);
                for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) Gr[(size_t)j * m + i] = G0[(size_t)j * ld + i].x;
                ratio = fh_pivoted_cholesky_ratio_real(std::move(Gr), m);
            } else {
                ratio = fh_pivoted_cholesky_ratio(std::move(G0), m, ld);
            }
            static const bool always_two = getenv("FH_CHOLQR_TWO_PASS") != nullptr;
            // the one-pass condition of fh_ortho_panel (pivot ratio of the equilibrated Gram matrix > 1e-2: the basis below is
            // then as good as an orthonormalised one to 1e-14) and the reference's rank rule with the same margin as there
            fast = ratio > 1e-2 && !always_two && (dmin / dmax) * std::sqrt(ratio) > 1e3 * rank_tol;
        }
    }
    if (fast) {
        // basis = Q_proj D^-1 (unit columns): its pencil is the equilibrated Gram pair; the orthonormal basis is never formed
        emit(Gh + g2, dcol.data(), m, Aq_host);
        emit(b_id ? Gh : Gh + 2 * g2, dcol.data(), m, Bq_host);
        h->rs_basis = P; h->rs_rank = m;
        h->rs_T.assign(m, cmake(1, 0));
        for (int j = 0; j < m; ++j) h->rs_T[j] = cmake(1.0 / dcol[j], 0.0);
        *rank = m;
        fh_prof_collect(h);
        FH_CHECK(hipGetLastError());
        return 0;
    }
    // ---- general path: the rank-revealing orthonormalisation on the resident panel, then the projections of its result ----
    if ((rc = fh_get_buf(h, "or_out", (size_t)N * FH_MAX_LD * sizeof(cplx), &p))) return rc;
    cplx* Out = (cplx*)p;
    cplx* res = nullptr;
    int r = 0;
    if ((rc = fh_ortho_panel(h, m, ld, P, Out, rank_tol, 0.0, m, &r, &res))) return rc;
    h->rs_P = nullptr;                            // (the orthonormalisation may have overwritten the projection panel)
    *rank = r;
    h->rs_rank = r; h->rs_T.clear(); h->rs_basis = res;
    if (r == 0) { fh_prof_collect(h); return 0; }
    if ((rc = grams(res, false))) return rc;
    emit(Gh + g2, nullptr, r, Aq_host);
    if (b_id) identity(r, Bq_host);               // orthonormal basis, B = I: exactly I (src/dense/feast_dense.jl:255-259)
    else emit(Gh + 2 * g2, nullptr, r, Bq_host);
    fh_prof_collect(h);
    FH_CHECK(hipGetLastError());
    return 0;
}

extern "C" int feasthip_rr_ritz_resident(feasthip_handle h, int64_t r64, const void* V_host, const double* lambda_host, int64_t M,
                                         int normalize, int use_B, double* res_host) {
    int rc = fh_check_problem(h, r64);
    if (rc) return rc;
    if (!V_host || !lambda_host) { h->last_error = "rr_ritz_resident: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    if (M < 0 || M > r64) { h->last_error = "rr_ritz_resident: M out of range"; return FEASTHIP_ERROR_M0; }
    FH_CHECK(hipSetDevice(h->device));
    const int r = (int)r64, N = (int)fh_N(h);
    cplx *Pb, *Xp, *Rp;
    if ((rc = fh_rs_panels(h, &Pb, &Xp, &Rp))) return rc;
    if (!h->rs_basis || h->rs_rank != r) { h->last_error = "rr_ritz_resident: run rr_reduce_resident first (rank mismatch)"; return FEASTHIP_ERROR_M0; }
    const int ld = h->rs_ld;
    void* p;
    if ((rc = fh_get_buf(h, "rz_V", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
    cplx* dV = (cplx*)p;
    const int nblk_op = fh_op_nblk(h, ld), nblk_vec = fh_vec_nblk(N, ld);
    if ((rc = fh_get_buf(h, "rz_part", (size_t)std::max(nblk_op, nblk_vec) * ld * sizeof(cplx), &p))) return rc;
    cplx* part = (cplx*)p;
    if ((rc = fh_get_buf(h, "rz_dots", (size_t)ld * sizeof(cplx), &p))) return rc;
    cplx* ddots = (cplx*)p;
    // V padded to ld x ld; the implicit basis is Q_proj D^-1, so X = Q_proj (D^-1 V)
    std::vector<cplx> Vp((size_t)ld * ld, cmake(0, 0));
    const cplx* Vh = (const cplx*)V_host;
    const bool scaled = !h->rs_T.empty();
    for (int j = 0; j < r; ++j)
        for (int i = 0; i < r; ++i) {
            const cplx v = Vh[(size_t)j * r + i];
            Vp[(size_t)j * ld + i] = scaled ? cscale(v, h->rs_T[i].x) : v;
        }
    if ((rc = fh_upload_small(h, dV, Vp.data(), Vp.size() * sizeof(cplx)))) return rc;
    h->rs_X = nullptr; h->rs_R = nullptr;
    fh_prof_begin(h, "ritz");
    fh_launch_small_matmul(h->rs_basis, dV, N, ld, Xp, h->stream);
    fh_prof_end(h);
    if (normalize && M > 0) {
        fh_launch_dot_cols(Xp, Xp, N, ld, part, ddots, h->stream);
        fh_launch_normalize_cols(Xp, ddots, N, ld, (int)M, h->stream);
    }
    // R = A X - B X diag(lambda) for all r columns (the next sweep's start residual); res_j = ||R_j|| / max(|lambda_j|, 1), j < M
    std::vector<cplx> ca(ld, cmake(1, 0)), cb(ld, cmake(0, 0));
    const bool lam_in_op = use_B || fh_b_identity(h);
    for (int c = 0; c < r; ++c) cb[c] = lam_in_op ? cmake(-lambda_host[2 * c], -lambda_host[2 * c + 1]) : cmake(0, 0);
    cplx *dca, *dcb;
    if ((rc = fh_upload_coefs(h, "rz_coefA", ca, &dca))) return rc;
    if ((rc = fh_upload_coefs(h, "rz_coefB", cb, &dcb))) return rc;
    fh_op_call oc;
    oc.m = r;
    oc.X = Xp; oc.x_stride = 0; oc.Y = Rp; oc.y_stride = 0; oc.coefA = dca; oc.coefB = dcb;
    oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
    oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
    fh_apply_operator(h, ld, oc);
    if (!lam_in_op) {
        // RCI-style residual without B: R = A X - X diag(lambda)  (src/kernel/feast_kernel.jl:899-906)
        std::vector<cplx> lam(ld, cmake(0, 0));
        for (int c = 0; c < r; ++c) lam[c] = cmake(lambda_host[2 * c], lambda_host[2 * c + 1]);
        cplx* dl;
        if ((rc = fh_upload_coefs(h, "rz_lam", lam, &dl))) return rc;
        fh_launch_axpy_cols(Rp, Xp, dl, N, ld, h->stream);
    }
    const cplx* dots_h = nullptr;
    std::vector<char> dots_fb;
    if (M > 0 && res_host) {
        fh_launch_dot_cols(Rp, Rp, N, ld, part, ddots, h->stream);
        const void* slot = nullptr;
        if ((rc = fh_download_small(h, ddots, ld * sizeof(cplx), &slot, dots_fb))) return rc;
        dots_h = (const cplx*)slot;
    }
    // the rank dropped below the panel's padded width (64 -> 32 / 16 columns): the next sweep works on the narrower panel
    int ldx = ld;
    if (fh_pick_ld(r) < ld) {
        ldx = fh_pick_ld(r);
        if ((rc = fh_get_buf(h, "rs_tmp", (size_t)N * FH_MAX_LD * sizeof(cplx), &p))) return rc;
        cplx* tmp = (cplx*)p;
        for (cplx* pan : {Xp, Rp}) {
            fh_launch_panel_cols(pan, ld, 0, r, N, tmp, ldx, h->stream);
            FH_CHECK(hipMemcpyAsync(pan, tmp, (size_t)N * ldx * sizeof(cplx), hipMemcpyDeviceToDevice, h->stream));
        }
    }
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (M > 0 && res_host)
        for (int c = 0; c < (int)M; ++c) {
            const double la = std::hypot(lambda_host[2 * c], lambda_host[2 * c + 1]);
            res_host[c] = std::sqrt(dots_h[c].x) / std::max(la, 1.0);
        }
    h->rs_X = Xp; h->rs_X_m = r; h->rs_X_ld = ldx;
    h->rs_R_lambda.assign(ld, cmake(0, 0));
    if (use_B || fh_b_identity(h)) {                  // the start residual of the sweeps is the one WITH B
        for (int c = 0; c < r; ++c) h->rs_R_lambda[c] = cmake(lambda_host[2 * c], lambda_host[2 * c + 1]);
        h->rs_R = Rp;
    }
    fh_prof_collect(h);
    FH_CHECK(hipGetLastError());
    return 0;
}

extern "C" int feasthip_resident_export(feasthip_handle h, int which, int64_t ncols, void* dX) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (h->poisoned) { h->last_error = "handle poisoned by an earlier device failure: destroy it"; return FEASTHIP_ERROR_INTERNAL; }
    if (!dX) { h->last_error = "resident_export: null destination"; return FEASTHIP_ERROR_INTERNAL; }
    const cplx* src = which == 0 ? h->rs_X : h->rs_P;
    const int have = which == 0 ? h->rs_X_m : h->rs_m, ld = which == 0 ? h->rs_X_ld : h->rs_ld;
    if (!src || ncols < 0 || ncols > have) { h->last_error = "resident_export: no such resident panel / too many columns"; return FEASTHIP_ERROR_M0; }
    FH_CHECK(hipSetDevice(h->device));
    const int N = (int)fh_N(h);
    if (ncols > 0) fh_launch_from_panel(src, ld, N, (int)ncols, (cplx*)dX, N, h->stream, fh_perm(h));
    FH_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int feasthip_resident_import(feasthip_handle h, int which, int64_t ncols, const void* dX) {
    int rc = fh_check_problem(h, ncols);
    if (rc) return rc;
    if (!dX) { h->last_error = "resident_import: null source"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const int m = (int)ncols, ld = fh_pick_ld(m), N = (int)fh_N(h);
    cplx *P, *X, *R;
    if ((rc = fh_rs_panels(h, &P, &X, &R))) return rc;
    fh_launch_to_panel((const cplx*)dX, N, N, m, which == 0 ? X : P, ld, h->stream, fh_perm(h));
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (which == 0) { h->rs_X = X; h->rs_X_m = m; h->rs_X_ld = ld; h->rs_R = nullptr; h->rs_R_lambda.clear(); }
    else { h->rs_P = P; h->rs_m = m; h->rs_ld = ld; h->rs_basis = nullptr; h->rs_T.clear(); h->rs_rank = 0; }
    return 0;
}

// ---------------------------------------------------------------------------------------
// Host policy of the inexact FEAST mode (fh_policy.hpp): exported so that host shims call it instead of re-porting it
// ---------------------------------------------------------------------------------------
extern "C" int feasthip_policy_init(feasthip_policy* p, double Emin, double Emax, int ne, int quadrature, double inner_rtol,
                                    double outer_tol, int solver_maxiter, int steer, int fpm18) {
    if (!p || !(Emax > Emin) || ne < 1 || !(inner_rtol > 0.0) || solver_maxiter < 1) return FEASTHIP_ERROR_FPM;
    memset(p, 0, sizeof(*p));
    p->Emin = Emin; p->Emax = Emax; p->inner_rtol = inner_rtol; p->outer_tol = outer_tol;
    p->ne = ne; p->quadrature = quadrature;
    p->steer = (steer && (quadrature == 0 || quadrature == 1)) ? 1 : 0;
    p->cap = 8000; p->inner_cap = p->base_cap = solver_maxiter;
    p->eps_prev = INFINITY; p->next_rtol = inner_rtol; p->last_reach = -1.0;
    // a priori the subspace (1.5 x the eigenvalue count is the usual M0) reaches about 1.4 half widths
    p->aspect = p->steer ? fh_policy::pick(Emin, Emax, ne, quadrature, inner_rtol, p->cap, 1.4, nullptr, 0, 0) : fpm18;
    return 0;
}

extern "C" int feasthip_policy_update(feasthip_policy* p, double epsout, int M, int any_node_capped, const double* ritz, int nritz) {
    if (!p || (nritz > 0 && !ritz) || M < 0 || M > nritz) return FEASTHIP_ERROR_FPM;
    // stagnation guard: the outer residual should contract by about inner_rtol per loop.  When it has not even halved over
    // two loops the inner solves are not delivering (iteration cap too low for this matrix): double the cap
    if (p->n_hist < 3) p->eps_hist[p->n_hist++] = epsout;
    else { p->eps_hist[0] = p->eps_hist[1]; p->eps_hist[1] = p->eps_hist[2]; p->eps_hist[2] = epsout; }
    if (p->n_hist >= 3 && p->eps_hist[2] > 0.5 * p->eps_hist[0] && p->inner_cap < 16 * p->base_cap) {
        p->inner_cap *= 2;
        p->n_hist = 0;
    }
    if (p->steer) {
        // Safeguard first.  The policy promised a contraction of max(filter ratio, inner_rtol) < 0.5 per loop.  When a loop
        // delivers less than 0.3 there are two possible culprits: inner solves that stopped at the iteration cap before
        // reaching inner_rtol (a taller ellipse would only HELP them -- raise the cap instead), or a filter that is too soft
        // for this spectrum (lower the ellipse, down to the circle).
        if (std::isfinite(p->eps_prev) && std::isfinite(epsout) && epsout > 0.3 * p->eps_prev) {
            if (any_node_capped && p->inner_cap < 16 * p->base_cap) { p->inner_cap *= 2; p->n_hist = 0; }
            else if (p->aspect > 100) p->cap = std::max(100, p->aspect / 2);
        }
        // Steering: the filter model at the reach of the subspace.  The guard Ritz values overshoot outward while they are far
        // from converged, hence a cautious quantile of their distances early, nearly the outermost one later, and never more
        // than double the ratio in one loop.
        const double reach = M > 0 ? fh_policy::subspace_reach(ritz, nritz, p->Emin, p->Emax, !(epsout < 1e-2) ? 0.8 : 0.95) : -1.0;
        p->last_reach = reach;
        p->aspect = reach >= 0.0 ? fh_policy::pick(p->Emin, p->Emax, p->ne, p->quadrature, p->inner_rtol, p->cap, reach, ritz, M, 2 * p->aspect)
                                 : std::min(p->aspect, p->cap);
    }
    // The last loop: a sweep reduces the outer residual by about 2 x its inner tolerance; when less than that is still
    // needed to reach outer_tol, the next sweep's inner tolerance is relaxed to what is needed (with a margin of 3), never
    // beyond 0.3 -- on cfg 3 a loop that starts at 1.3e-12 for a target of 1e-12 costs a third of a full one
    p->next_rtol = p->inner_rtol;
    if (p->outer_tol > 0.0 && std::isfinite(epsout) && epsout > p->outer_tol)
        p->next_rtol = std::min(0.3, std::max(p->inner_rtol, 0.32 * p->outer_tol / epsout));
    p->eps_prev = epsout;
    return 0;
}

extern "C" int feasthip_policy_set_aside(const double* res, int M, int* flags) {
    // Inexact inner solves leave solver noise in the guard columns.  Its Ritz values are arbitrary; one that lands inside
    // the interval has an O(1) residual that never contracts and would hold epsout up forever (variant A has no
    // spurious-pair removal; with exact solves the guard columns are true eigen-directions and stay outside).  A pair is set
    // aside when its relative residual is > 0.1 AND > 100x the smallest residual of the pairs inside: a true pair inside the
    // interval sees a filter value >= 1/2 and contracts with the others.
    if (!res || !flags || M <= 1) { if (flags) for (int j = 0; j < M; ++j) flags[j] = 0; return 0; }
    double rmin = res[0];
    for (int j = 1; j < M; ++j) rmin = std::min(rmin, res[j]);
    int n = 0;
    for (int j = 0; j < M; ++j) { flags[j] = (res[j] > 0.1 && res[j] > 100.0 * rmin) ? 1 : 0; n += flags[j]; }
    if (n == 0 || n >= M) { for (int j = 0; j < M; ++j) flags[j] = 0; return 0; }
    return n;
}

extern "C" double feasthip_policy_filter_ratio(double Emin, double Emax, int ne, int quadrature, int fpm18, double reach,
                                               const double* inside, int n_inside) {
    if (!(Emax > Emin) || ne < 1 || fpm18 < 0 || (quadrature != 0 && quadrature != 1)) return NAN;
    return fh_policy::filter_ratio(Emin, Emax, ne, quadrature, fpm18, reach, inside, n_inside);
}

extern "C" double feasthip_policy_reach(const double* ritz, int n, double Emin, double Emax, double quantile) {
    if (n > 0 && !ritz) return -1.0;
    return fh_policy::subspace_reach(ritz, n, Emin, Emax, quantile);
}

// ---------------------------------------------------------------------------------------
// RCI seams: Y = A X / B X (jobs 30/40), Y = (zB - A)^{-1} X (jobs 10+11, linear_solver)
// ---------------------------------------------------------------------------------------
// Rayleigh-Ritz step with the reduced eigenproblem on the device: project,
// Jacobi eigensolver (fh_eig.hip) instead of host ZHEGV, stable inside-first reorder for [Emin, Emax]
// (src/core/feast_aux.jl:144-197), X = Q V, normalise the inside columns, residuals.  The host sees
// lambda[r] (reordered), M and res[M] only.  FEASTHIP_ERROR_LAPACK: the reduced B matrix is not
// positive definite -- the caller falls back to project + host eigen + ritz_residual
// (the general fallback of src/dense/feast_dense.jl:276-284).
extern "C" int feasthip_rayleigh_ritz_dev(feasthip_handle h, int64_t r64, const void* dQ, double Emin, double Emax,
                                          int use_B, void* dX, double* lambda_out, int* M_out, double* res_out) {
    int rc = fh_check_problem(h, r64);
    if (rc) return rc;
    if (!dQ || !dX || !lambda_out || !M_out) { h->last_error = "rayleigh_ritz: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    const int r = (int)r64, ld = FH_MAX_LD;
    std::vector<cplx> Sq((size_t)r * r), Aq((size_t)r * r);
    if ((rc = feasthip_project_dev(h, r64, dQ, 0, 1, Sq.data(), Aq.data()))) return rc;
    bool a_identity = fh_b_identity(h) != 0;                 // project returned exactly I
    void* p;
    if ((rc = fh_get_buf(h, "rr_S", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
    cplx* dS = (cplx*)p;
    if ((rc = fh_get_buf(h, "rr_A", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
    cplx* dA = (cplx*)p;
    if ((rc = fh_get_buf(h, "rr_V", (size_t)ld * ld * sizeof(cplx), &p))) return rc;
    cplx* dV = (cplx*)p;
    if ((rc = fh_get_buf(h, "rr_lam", ld * sizeof(double), &p))) return rc;
    double* dlam = (double*)p;
    if ((rc = fh_get_buf(h, "rr_flags", 4 * sizeof(int), &p))) return rc;
    int* dflags = (int*)p;
    if ((rc = fh_get_buf(h, "rr_scratch", fh_herm_eig_scratch_bytes(), &p))) return rc;
    void* scratch = p;
    std::vector<cplx> pad((size_t)ld * ld, cmake(0, 0));
    for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) pad[(size_t)j * ld + i] = Sq[(size_t)j * r + i];
    FH_CHECK(hipMemcpyAsync(dS, pad.data(), pad.size() * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (!a_identity) {
        for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) pad[(size_t)j * ld + i] = Aq[(size_t)j * r + i];
        FH_CHECK(hipMemcpyAsync(dA, pad.data(), pad.size() * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
        FH_CHECK(hipStreamSynchronize(h->stream));
    }
    fh_prof_begin(h, "reduced_eig");
    fh_launch_herm_eig(r, ld, dS, a_identity ? nullptr : dA, scratch, dlam, dV, dflags, h->stream);
    fh_prof_end(h);
    std::vector<double> lam(r);
    std::vector<cplx> V((size_t)ld * ld);
    int flags[4];
    FH_CHECK(hipMemcpyAsync(flags, dflags, sizeof(flags), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipMemcpyAsync(lam.data(), dlam, r * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipMemcpyAsync(V.data(), dV, V.size() * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
    FH_CHECK(hipStreamSynchronize(h->stream));
    if (getenv("FH_DEBUG_TIMING")) fprintf(stderr, "[rayleigh_ritz] r=%d jacobi sweeps=%d\n", r, flags[3]);
    if (flags[0] || flags[2]) { h->last_error = "rayleigh_ritz: reduced B matrix not positive definite"; return FEASTHIP_ERROR_LAPACK; }
    // stable inside-first permutation
    std::vector<int> perm;
    for (int i = 0; i < r; ++i) if (lam[i] >= Emin && lam[i] <= Emax) perm.push_back(i);
    const int M = (int)perm.size();
    for (int i = 0; i < r; ++i) if (!(lam[i] >= Emin && lam[i] <= Emax)) perm.push_back(i);
    std::vector<cplx> Vs((size_t)r * r);
    std::vector<double> lamc(2 * (size_t)r);
    for (int k = 0; k < r; ++k) {
        lambda_out[k] = lam[perm[k]];
        lamc[2 * k] = lam[perm[k]]; lamc[2 * k + 1] = 0.0;
        for (int i = 0; i < r; ++i) Vs[(size_t)k * r + i] = V[(size_t)perm[k] * ld + i];
    }
    *M_out = M;
    return feasthip_ritz_residual_dev(h, r64, dQ, Vs.data(), lamc.data(), M, 1, use_B, dX, res_out);
}

extern "C" int feasthip_matmul_dev(feasthip_handle h, int which, int64_t m64, const void* dX, void* dY) {
    if (m64 > FH_MAX_LD) {          // independent columns: 64 at a time
        int rc0 = fh_check_problem(h, m64, 1);
        if (rc0) return rc0;
        const size_t N0 = (size_t)fh_N(h);
        for (int64_t c0 = 0; c0 < m64; c0 += FH_MAX_LD) {
            const int64_t mc = std::min<int64_t>(FH_MAX_LD, m64 - c0);
            rc0 = feasthip_matmul_dev(h, which, mc, (const cplx*)dX + c0 * N0, (cplx*)dY + c0 * N0);
            if (rc0) return rc0;
        }
        return 0;
    }

    int rc = fh_check_problem(h, m64);
    if (rc) return rc;
    if (!dX || !dY || (which != 0 && which != 1)) { h->last_error = "matmul: bad argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const int m = (int)m64, ld = fh_pick_ld(m), N = (int)fh_N(h);
    const size_t panel = (size_t)N * ld;
    void* p;
    if ((rc = fh_get_buf(h, "mm_X", panel * sizeof(cplx), &p))) return rc;
    cplx* Xp = (cplx*)p;
    if ((rc = fh_get_buf(h, "mm_Y", panel * sizeof(cplx), &p))) return rc;
    cplx* Yp = (cplx*)p;
    fh_launch_to_panel((const cplx*)dX, N, N, m, Xp, ld, h->stream, fh_perm(h));
    std::vector<cplx> ca(ld, cmake(which == 0 ? 1 : 0, 0)), cb(ld, cmake(which == 1 ? 1 : 0, 0));
    cplx *dca, *dcb;
    if ((rc = fh_upload_coefs(h, "mm_coefA", ca, &dca))) return rc;
    if ((rc = fh_upload_coefs(h, "mm_coefB", cb, &dcb))) return rc;
    fh_op_call oc;
    oc.m = m;
    oc.X = Xp; oc.x_stride = 0; oc.Y = Yp; oc.y_stride = 0; oc.coefA = dca; oc.coefB = dcb;
    oc.Bvec = nullptr; oc.b_stride = 0; oc.U = nullptr; oc.u_stride = 0; oc.dot_mode = 0;
    oc.partial1 = nullptr; oc.partial2 = nullptr; oc.node_active = nullptr; oc.nodes = 1;
    fh_apply_operator(h, ld, oc);
    fh_launch_from_panel(Yp, ld, N, m, (cplx*)dY, N, h->stream, fh_perm(h));
    FH_CHECK(hipStreamSynchronize(h->stream));
    fh_prof_collect(h);
    FH_CHECK(hipGetLastError());       // launch-configuration errors do not surface through the stream sync
    return 0;
}

extern "C" int feasthip_matmul(feasthip_handle h, int which, int64_t m, const void* X, void* Y) {
    int rc = fh_check_problem(h, m, 1);
    if (rc) return rc;
    if (!X || !Y) { h->last_error = "matmul: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const size_t nb = (size_t)fh_N(h) * m * sizeof(cplx);
    void *dX, *dY;
    if ((rc = fh_stage_in(h, "host_Q", X, nb, &dX))) return rc;
    if ((rc = fh_get_buf(h, "host_X", nb, &dY))) return rc;
    rc = feasthip_matmul_dev(h, which, m, dX, dY);
    if (rc) return rc;
    FH_CHECK(hipMemcpy(Y, dY, nb, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int feasthip_shifted_solve_dev(feasthip_handle h, double z_re, double z_im, int64_t m64, const void* dX,
                                          void* dY, feasthip_stats* stats) {
    if (m64 > FH_MAX_LD) {          // independent right-hand sides: 64 at a time (LU factor cached between panels)
        int rc0 = fh_check_problem(h, m64, 1);
        if (rc0) return rc0;
        const size_t N0 = (size_t)fh_N(h);
        feasthip_stats tot;
        memset(&tot, 0, sizeof(tot));
        int worst = 0;
        for (int64_t c0 = 0; c0 < m64; c0 += FH_MAX_LD) {
            const int64_t mc = std::min<int64_t>(FH_MAX_LD, m64 - c0);
            feasthip_stats st;
            rc0 = feasthip_shifted_solve_dev(h, z_re, z_im, mc, (const cplx*)dX + c0 * N0, (cplx*)dY + c0 * N0, &st);
            if (rc0 != 0 && rc0 != FEASTHIP_ERROR_NO_CONVERGENCE && rc0 != FEASTHIP_ERROR_LAPACK) return rc0;
            worst = std::max(worst, rc0);
            tot.krylov_iterations += st.krylov_iterations; tot.spmm_calls += st.spmm_calls; tot.factorizations += st.factorizations;
            tot.max_rel_residual = std::max(tot.max_rel_residual, st.max_rel_residual);
        }
        if (stats) *stats = tot;
        return worst;
    }

    int rc = fh_check_problem(h, m64);
    if (rc) return rc;
    if (!dX || !dY) { h->last_error = "shifted_solve: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const int m = (int)m64, ld = fh_pick_ld(m), N = (int)fh_N(h);
    const size_t panel = (size_t)N * ld;
    void* p;
    if ((rc = fh_get_buf(h, "ss_rhs", panel * sizeof(cplx), &p))) return rc;
    cplx* Rhs = (cplx*)p;
    if ((rc = fh_get_buf(h, "ss_Y", panel * sizeof(cplx), &p))) return rc;
    cplx* Y = (cplx*)p;
    if (stats) memset(stats, 0, sizeof(*stats));
    fh_launch_to_panel((const cplx*)dX, N, N, m, Rhs, ld, h->stream, fh_perm(h));
    std::vector<cplx> z(1, cmake(z_re, z_im));
    std::vector<int> status(1, 0);
    if (h->solver == FEASTHIP_SOLVER_LU) {
        if (h->kind != 1) { h->last_error = "solver LU requires a dense matrix"; return FEASTHIP_ERROR_FPM; }
        int64_t nfact = 0;
        // cached per quadrature node when z is one, else in one extra slot
        double worst = 0.0;
        if (h->factor_precision == 32) rc = fh_dense_lu_refined(h, ld, m, 1, z, Rhs, Y, panel, status, &nfact, true, &worst);
        else rc = fh_dense_lu_solve_single(h, ld, m, z[0], Rhs, Y, &status[0], &nfact);
        if (rc) return rc;
        if (stats) { stats->factorizations = nfact; stats->max_rel_residual = worst; }
    } else if (h->solver == FEASTHIP_SOLVER_BANDED) {
        int64_t nfact = 0;
        double worst = 0.0;
        if (h->factor_precision == 32) rc = fh_dense_lu_refined(h, ld, m, 1, z, Rhs, Y, panel, status, &nfact, true, &worst, true);
        else rc = fh_banded_solve_single(h, ld, m, z[0], Rhs, Y, &status[0], &nfact);
        if (rc) return rc;
        if (stats) { stats->factorizations = nfact; stats->max_rel_residual = worst; }
    } else if (h->solver == FEASTHIP_SOLVER_BICGSTAB || h->solver == FEASTHIP_SOLVER_COCG) {
        if (h->solver == FEASTHIP_SOLVER_COCG && fh_is_complex_input(h)) {
            h->last_error = "solver COCG needs real-symmetric A and B";
            return FEASTHIP_ERROR_FPM;
        }
        FH_CHECK(hipMemsetAsync(Y, 0, panel * sizeof(cplx), h->stream));
        fh_solve_result sr;
        rc = fh_krylov(h, h->solver == FEASTHIP_SOLVER_COCG ? 1 : 0, h->factor_precision, ld, m, 1, z, Rhs, Y, panel, sr);
        if (rc) return rc;
        status = sr.status;
        if (stats) { stats->krylov_iterations = sr.iters_sum; stats->spmm_calls = sr.op_calls; stats->max_rel_residual = sr.max_rel_res; }
    } else {
        FH_CHECK(hipMemsetAsync(Y, 0, panel * sizeof(cplx), h->stream));
        fh_solve_result sr;
        rc = fh_gmres(h, ld, m, 1, z, Rhs, Y, panel, sr);
        if (rc) return rc;
        status = sr.status;
        if (stats) { stats->krylov_iterations = sr.iters_sum; stats->spmm_calls = sr.op_calls; stats->max_rel_residual = sr.max_rel_res; }
    }
    fh_launch_from_panel(Y, ld, N, m, (cplx*)dY, N, h->stream, fh_perm(h));
    FH_CHECK(hipStreamSynchronize(h->stream));
    fh_prof_collect(h);
    return status[0];
}

extern "C" int feasthip_shifted_solve(feasthip_handle h, double z_re, double z_im, int64_t m, const void* X, void* Y,
                                      feasthip_stats* stats) {
    int rc = fh_check_problem(h, m, 1);
    if (rc) return rc;
    if (!X || !Y) { h->last_error = "shifted_solve: null argument"; return FEASTHIP_ERROR_INTERNAL; }
    FH_CHECK(hipSetDevice(h->device));
    const size_t nb = (size_t)fh_N(h) * m * sizeof(cplx);
    void *dX, *dY;
    if ((rc = fh_stage_in(h, "host_Q", X, nb, &dX))) return rc;
    if ((rc = fh_get_buf(h, "host_X", nb, &dY))) return rc;
    rc = feasthip_shifted_solve_dev(h, z_re, z_im, m, dX, dY, stats);
    if (rc != 0 && rc != FEASTHIP_ERROR_NO_CONVERGENCE) return rc;
    FH_CHECK(hipMemcpy(Y, dY, nb, hipMemcpyDeviceToHost));
    return rc;
}

// ---------------------------------------------------------------------------------------
// measurement support
// ---------------------------------------------------------------------------------------
extern "C" int feasthip_last_node_iterations(feasthip_handle h, int* out, int n) {
    if (!h || !out) return FEASTHIP_ERROR_INTERNAL;
    for (int e = 0; e < n; ++e) out[e] = e < (int)h->last_node_iters.size() ? h->last_node_iters[e] : 0;
    return 0;
}

extern "C" int feasthip_last_global_node_iterations(feasthip_handle h, int* out, int n) {
    if (!h || !out) return FEASTHIP_ERROR_INTERNAL;
    for (int e = 0; e < n; ++e) out[e] = e < (int)h->global_node_iters.size() ? h->global_node_iters[e] : 0;
    return 0;
}

extern "C" int feasthip_last_column_iterations(feasthip_handle h, int* out, int n) {
    if (!h || !out) return FEASTHIP_ERROR_INTERNAL;
    for (int e = 0; e < n; ++e) out[e] = e < (int)h->last_col_iters.size() ? h->last_col_iters[e] : 0;
    return 0;
}

extern "C" int feasthip_profile_enable(feasthip_handle h, int enable) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    h->profiling = enable ? 1 : 0;
    if (enable) { h->prof_host_s = 0.0; h->prof_t0 = fh_now_s(); }
    return 0;
}
extern "C" int feasthip_profile_reset(feasthip_handle h) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    fh_prof_collect(h);
    h->prof.clear();
    h->prof_work.clear();
    h->prof_mult = 1;
    hipMemset(h->d_counters, 0, 8 * sizeof(unsigned long long));
    return 0;
}
extern "C" int feasthip_profile_set_period(feasthip_handle h, int period) {
    if (!h || period < 0) return FEASTHIP_ERROR_INTERNAL;
    h->prof_period = period;
    return 0;
}
extern "C" int feasthip_profile_get_work(feasthip_handle h, const char* kernel_class, double* work) {
    if (!h || !kernel_class || !work) return FEASTHIP_ERROR_INTERNAL;
    auto it = h->prof_work.find(kernel_class);
    *work = it == h->prof_work.end() ? 0.0 : it->second;
    return 0;
}
extern "C" int feasthip_profile_get(feasthip_handle h, const char* kernel_class, double* total_ms, int64_t* launches) {
    if (!h || !kernel_class) return FEASTHIP_ERROR_INTERNAL;
    fh_prof_collect(h);
    // device-side work counters: [0] active node-sweeps of the SpMM, [1] its active column x vector passes, [2] columns that
    // took a step in an update kernel, [3] of those the columns that go on iterating (the fused vector kernel reads and
    // writes five panels for them, one for a column on its last step), [4] distinct columns whose accumulator entries a
    // sum-mode launch read and wrote, [5] the part of [3] counted in the first vector launch of a lazy start (four passes)
    static const char* const names[] = {"spmm.node_launches", "spmm.column_passes", "update.active_columns", "update.continuing_columns",
                                        "update.accumulator_columns", "update.first_launch_columns"};
    for (int k = 0; k < 6; ++k)
        if (!strcmp(kernel_class, names[k])) {
            unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            hipStreamSynchronize(h->stream);
            hipMemcpy(c, h->d_counters, sizeof(c), hipMemcpyDeviceToHost);
            if (launches) *launches = (int64_t)c[k];
            if (total_ms) *total_ms = 0.0;
            return 0;
        }
    auto it = h->prof.find(kernel_class);
    auto is = h->prof.find(std::string(kernel_class) + "#sampled");
    int64_t n = it == h->prof.end() ? 0 : it->second.launches;
    double avg = 0.0;
    if (is != h->prof.end() && is->second.launches > 0) avg = is->second.total_ms / (double)is->second.launches;
    if (total_ms) *total_ms = avg * (double)n;   // estimated total = sampled average x launches
    if (launches) *launches = n;
    return 0;
}
