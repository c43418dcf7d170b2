// fh_comm.hip -- the per-loop reduction of the :hip backend inside the C ABI.
//
// Replaces the reductions the reference's parallel backends perform themselves:
//   MPI.Allreduce(local_Aq / local_Sq / local_Q, +, comm)   src/parallel/feast_mpi.jl:117-119, 856-858, 1001
//   master sum over per-node worker results                  src/parallel/feast_parallel.jl:497-503
//
// Transport 1 (default): RCCL.  librccl is resolved at run time (dlopen), so a host that already carries an
// RCCL (PyTorch bundles one) shares it and a Julia host picks up /opt/rocm/lib/librccl.so; one rank per GPU,
// ncclAllReduce(SUM, f64) on the handle's stream over xGMI.
// Transport 2 ("shm"): ranks that SHARE a device (RCCL refuses two ranks of one communicator on one HIP
// device: "Duplicate GPU detected"), used by test rigs that rehearse N ranks on one card -- as processes, as threads of one
// process (one handle per thread), or a mix.  Rendezvous through a POSIX shared-memory segment named after the unique id;
// every rank exports a staging buffer with hipIpcGetMemHandle, peers of other processes map it (peers of the same process
// take its pointer), and a kernel sums the staged buffers in RANK ORDER (bitwise identical on every rank).  Host-synchronous
// by design; it is a rehearsal transport, not the fast path.
#include "fh_common.hpp"
#include "fh_comm.hpp"
#include "../../include/feasthip.h"

#include <rccl/rccl.h>       // types only: the library itself is resolved with dlopen at run time

#include <atomic>
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <random>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

// ---- RCCL entry points, resolved lazily ----------------------------------------------------
namespace {
struct rccl_api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

rccl_api* rccl() {
    static rccl_api api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    // an RCCL that is already mapped into the process (e.g. torch/lib/librccl.so) wins: two RCCL copies in
    // one process would each run their own bootstrap and proxy threads
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (api.lib) break;
    }
    if (!api.lib) {
        const char* env = getenv("FEASTHIP_RCCL_LIB");
        if (env) api.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        for (size_t i = 0; !api.lib && i < sizeof(names) / sizeof(names[0]); ++i) api.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    }
    if (!api.lib) { api.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return &api; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.CommAbort = (decltype(api.CommAbort))dlsym(api.lib, "ncclCommAbort");       // optional
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) {
        api.error = "librccl lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllReduce";
        api.lib = nullptr;
    }
    return &api;
}

// ---- shared-memory rendezvous of the "shm" transport ---------------------------------------
constexpr int FH_SHM_MAX_RANKS = 16;
struct shm_segment {
    std::atomic<unsigned> attached;                  // ranks that mapped the segment
    std::atomic<unsigned> barrier_count;
    std::atomic<unsigned> barrier_sense;
    std::atomic<unsigned> failed;                    // a rank hit an error: everybody leaves the barrier
    hipIpcMemHandle_t handle[FH_SHM_MAX_RANKS];
    unsigned long long bytes[FH_SHM_MAX_RANKS];
    // ranks may also be THREADS of one process (one handle per thread): a peer of the same process is reached through
    // its device pointer, hipIpcOpenMemHandle refuses a handle exported by the opening process itself
    long long pid[FH_SHM_MAX_RANKS];
    unsigned long long ptr[FH_SHM_MAX_RANKS];
};

uint64_t fnv1a(const char* p, size_t n) {
    uint64_t hsh = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { hsh ^= (unsigned char)p[i]; hsh *= 1099511628211ull; }
    return hsh;
}
}   // namespace

struct fh_comm {
    int nranks = 1, rank = 0, transport = 0;
    ncclComm_t nccl = nullptr;
    // shm transport
    shm_segment* seg = nullptr;
    std::string shm_name;
    unsigned sense = 0;
    void* staging = nullptr; size_t staging_bytes = 0;
    void* peer[FH_SHM_MAX_RANKS] = {nullptr};
    bool peer_ipc[FH_SHM_MAX_RANKS] = {false};       // mapped with hipIpcOpenMemHandle (to be closed), not a same-process pointer
    const double** d_peers = nullptr;                // device array of the mapped peer pointers
    double timeout_s = 120.0;
};

// sense-reversing barrier in host shared memory; false on timeout or when a peer reported failure
static bool shm_barrier(fh_comm* c) {
    shm_segment* s = c->seg;
    c->sense ^= 1u;
    if (s->barrier_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)c->nranks) {
        s->barrier_count.store(0, std::memory_order_relaxed);
        s->barrier_sense.store(c->sense, std::memory_order_release);
        return s->failed.load() == 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; s->barrier_sense.load(std::memory_order_acquire) != c->sense; ++spins) {
        if (s->failed.load()) return false;
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
            s->failed.store(1);
            return false;
        }
    }
    return s->failed.load() == 0;
}

// out[i] = sum_r peers[r][i], ranks in index order (deterministic, identical on every rank)
__global__ __launch_bounds__(256) void k_sum_peers(const double* const* __restrict__ peers, int nranks, double* __restrict__ out, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < nranks; ++r) s += peers[r][i];
        out[i] = s;
    }
}

static void shm_unmap_peers(fh_comm* c) {
    for (int r = 0; r < c->nranks; ++r) {
        if (c->peer[r] && c->peer_ipc[r]) hipIpcCloseMemHandle(c->peer[r]);
        c->peer[r] = nullptr; c->peer_ipc[r] = false;
    }
}

// One staging buffer per rank, allocated and exchanged ONCE at init (re-exporting a re-allocated buffer was seen to
// fail with hipIpcGetMemHandle: invalid argument); larger reductions stream through it in chunks.
static int shm_setup_staging(feasthip_ctx* h, fh_comm* c) {
    auto fail = [&](const std::string& msg) { c->seg->failed.store(1); h->last_error = msg; return (int)FEASTHIP_ERROR_INTERNAL; };
    size_t cap = 32u << 20;
    if (getenv("FEASTHIP_COMM_STAGING_MB")) cap = (size_t)std::max(1, atoi(getenv("FEASTHIP_COMM_STAGING_MB"))) << 20;
    hipError_t e = hipMalloc(&c->staging, cap);
    if (e != hipSuccess) return fail(std::string("comm(shm): hipMalloc(staging): ") + hipGetErrorString(e));
    c->staging_bytes = cap;
    e = hipIpcGetMemHandle(&c->seg->handle[c->rank], c->staging);
    if (e != hipSuccess) return fail(std::string("comm(shm): hipIpcGetMemHandle: ") + hipGetErrorString(e) +
                                     " (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set for dmabuf IPC)");
    c->seg->bytes[c->rank] = cap;
    c->seg->pid[c->rank] = (long long)getpid();
    c->seg->ptr[c->rank] = (unsigned long long)(uintptr_t)c->staging;
    if (!shm_barrier(c)) { h->last_error = "comm(shm): peers did not publish their staging buffers"; return FEASTHIP_ERROR_INTERNAL; }
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) { c->peer[r] = c->staging; continue; }
        if (c->seg->pid[r] == (long long)getpid()) {
            c->peer[r] = (void*)(uintptr_t)c->seg->ptr[r];        // a thread of this process: same address space
        } else {
            e = hipIpcOpenMemHandle(&c->peer[r], c->seg->handle[r], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) return fail(std::string("comm(shm): hipIpcOpenMemHandle: ") + hipGetErrorString(e));
            c->peer_ipc[r] = true;
        }
        c->staging_bytes = std::min<size_t>(c->staging_bytes, (size_t)c->seg->bytes[r]);
    }
    e = hipMalloc((void**)&c->d_peers, FH_SHM_MAX_RANKS * sizeof(double*));
    if (e == hipSuccess) e = hipMemcpy(c->d_peers, c->peer, c->nranks * sizeof(double*), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(std::string("comm(shm): peer table: ") + hipGetErrorString(e));
    if (!shm_barrier(c)) { h->last_error = "comm(shm): peers did not map the staging buffers"; return FEASTHIP_ERROR_INTERNAL; }
    return 0;
}

// ---------------------------------------------------------------------------------------------
extern "C" int feasthip_comm_unique_id(char* uid) {
    if (!uid) return FEASTHIP_ERROR_INTERNAL;
    memset(uid, 0, FEASTHIP_UNIQUE_ID_BYTES);
    rccl_api* r = rccl();
    if (r->lib) {
        ncclUniqueId id;
        if (r->GetUniqueId(&id) == ncclSuccess) { memcpy(uid, id.internal, NCCL_UNIQUE_ID_BYTES); return 0; }
    }
    // no RCCL in this process: a random id still serves the shm transport
    std::random_device rd;
    for (int i = 0; i < FEASTHIP_UNIQUE_ID_BYTES; i += 4) { unsigned v = rd(); memcpy(uid + i, &v, 4); }
    return 0;
}

int fh_comm_destroy(feasthip_ctx* h) {
    fh_comm* c = h->comm;
    if (!c) return 0;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (c->transport == FEASTHIP_COMM_RCCL && c->nccl) rccl()->CommDestroy(c->nccl);
    if (c->transport == FEASTHIP_COMM_SHM && c->seg) {
        if (!c->seg->failed.load()) shm_barrier(c);  // nobody unmaps while a peer may still read
        shm_unmap_peers(c);
        if (c->staging) hipFree(c->staging);
        if (c->d_peers) hipFree((void*)c->d_peers);
        munmap(c->seg, sizeof(shm_segment));
    }
    delete c;
    h->comm = nullptr;
    return 0;
}

extern "C" int feasthip_comm_destroy(feasthip_handle h) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    return fh_comm_destroy(h);
}

extern "C" int feasthip_comm_init_rank(feasthip_handle h, int nranks, int rank, const char* uid, int transport) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (nranks < 1 || rank < 0 || rank >= nranks || !uid) { h->last_error = "comm_init_rank: need 0 <= rank < nranks and a unique id"; return FEASTHIP_ERROR_INTERNAL; }
    fh_comm_destroy(h);
    FH_CHECK(hipSetDevice(h->device));
    if (transport == FEASTHIP_COMM_AUTO) {
        const char* env = getenv("FEASTHIP_COMM_TRANSPORT");
        transport = (env && !strcmp(env, "shm")) ? FEASTHIP_COMM_SHM : FEASTHIP_COMM_RCCL;
    }
    fh_comm* c = new fh_comm();
    c->nranks = nranks; c->rank = rank; c->transport = transport;
    if (getenv("FEASTHIP_COMM_TIMEOUT_S")) c->timeout_s = std::max(1.0, atof(getenv("FEASTHIP_COMM_TIMEOUT_S")));
    if (transport == FEASTHIP_COMM_RCCL) {
        rccl_api* r = rccl();
        if (!r->lib) { h->last_error = "comm_init_rank: " + r->error; delete c; return FEASTHIP_ERROR_INTERNAL; }
        ncclUniqueId id;
        memcpy(id.internal, uid, NCCL_UNIQUE_ID_BYTES);
        ncclResult_t e = r->CommInitRank(&c->nccl, nranks, id, rank);
        if (e != ncclSuccess) {
            h->last_error = std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(e) : "error") +
                            " (two ranks on one HIP device need transport = FEASTHIP_COMM_SHM)";
            delete c;
            return FEASTHIP_ERROR_INTERNAL;
        }
    } else if (transport == FEASTHIP_COMM_SHM) {
        if (nranks > FH_SHM_MAX_RANKS) { h->last_error = "comm(shm): at most 16 ranks"; delete c; return FEASTHIP_ERROR_INTERNAL; }
        char name[64];
        snprintf(name, sizeof(name), "/feasthip_%016llx", (unsigned long long)fnv1a(uid, FEASTHIP_UNIQUE_ID_BYTES));
        c->shm_name = name;
        int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(shm_segment)) != 0) {
            h->last_error = std::string("comm(shm): shm_open/ftruncate ") + name + " failed";
            if (fd >= 0) close(fd);
            delete c;
            return FEASTHIP_ERROR_INTERNAL;
        }
        void* p = mmap(nullptr, sizeof(shm_segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { h->last_error = "comm(shm): mmap failed"; delete c; return FEASTHIP_ERROR_INTERNAL; }
        c->seg = (shm_segment*)p;                      // a fresh segment is zero-filled: counters start at 0
        // wait until everybody has mapped the segment, then remove the name (the mapping stays valid)
        c->seg->attached.fetch_add(1);
        const auto t0 = std::chrono::steady_clock::now();
        while (c->seg->attached.load() < (unsigned)nranks) {
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
                h->last_error = "comm(shm): timed out waiting for the other ranks to attach";
                munmap(p, sizeof(shm_segment)); shm_unlink(name); delete c;
                return FEASTHIP_ERROR_INTERNAL;
            }
        }
        h->comm = c;
        if (!shm_barrier(c)) { h->last_error = "comm(shm): attach barrier failed"; return FEASTHIP_ERROR_INTERNAL; }
        if (rank == 0) shm_unlink(name);
        int rc = shm_setup_staging(h, c);
        if (rc) { fh_comm_destroy(h); return rc; }
        return 0;
    } else {
        h->last_error = "comm_init_rank: transport must be FEASTHIP_COMM_AUTO, _RCCL or _SHM";
        delete c;
        return FEASTHIP_ERROR_INTERNAL;
    }
    h->comm = c;
    return 0;
}

extern "C" int feasthip_comm_info(feasthip_handle h, int* nranks, int* rank, int* transport) {
    if (!h) return FEASTHIP_ERROR_INTERNAL;
    if (nranks) *nranks = h->comm ? h->comm->nranks : 1;
    if (rank) *rank = h->comm ? h->comm->rank : 0;
    if (transport) *transport = h->comm ? h->comm->transport : 0;
    return 0;
}

int fh_comm_nranks(feasthip_ctx* h) { return h->comm ? h->comm->nranks : 1; }
int fh_comm_rank(feasthip_ctx* h) { return h->comm ? h->comm->rank : 0; }

void fh_comm_mark_failed(feasthip_ctx* h) {
    fh_comm* c = h->comm;
    if (!c || c->nranks == 1) return;
    if (c->transport == FEASTHIP_COMM_RCCL) {
        if (c->nccl && rccl()->CommAbort) { rccl()->CommAbort(c->nccl); c->nccl = nullptr; }
    } else if (c->seg) {
        c->seg->failed.store(1);
    }
}

// In-place sum over the ranks of `count` doubles at device pointer d, ordered on the handle's stream.
int fh_comm_allreduce_sum(feasthip_ctx* h, double* d, size_t count) {
    fh_comm* c = h->comm;
    if (!c || c->nranks == 1 || count == 0) return 0;
    if (c->transport == FEASTHIP_COMM_RCCL) {
        if (!c->nccl) { h->last_error = "comm(rccl): the communicator was aborted after an earlier failure"; return FEASTHIP_ERROR_INTERNAL; }
        ncclResult_t e = rccl()->AllReduce(d, d, count, ncclDouble, ncclSum, c->nccl, h->stream);
        if (e != ncclSuccess) {
            h->last_error = std::string("ncclAllReduce: ") + (rccl()->GetErrorString ? rccl()->GetErrorString(e) : "error");
            return FEASTHIP_ERROR_INTERNAL;
        }
        return 0;
    }
    // stream the buffer through the staging area in chunks: stage -> barrier -> every rank sums the peers' chunks in
    // rank order -> barrier (nobody overwrites its staging while a peer may still read it)
    auto fail = [&](const char* msg) { c->seg->failed.store(1); h->last_error = msg; return (int)FEASTHIP_ERROR_INTERNAL; };
    const size_t chunk = c->staging_bytes / sizeof(double);
    for (size_t off = 0; off < count; off += chunk) {
        const size_t n = std::min(chunk, count - off);
        if (hipMemcpyAsync(c->staging, d + off, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess)
            return fail("comm(shm): staging copy failed");
        if (!shm_barrier(c)) { h->last_error = "comm(shm): a peer failed or timed out before the reduction"; return FEASTHIP_ERROR_INTERNAL; }
        const int nblk = (int)std::min<size_t>(2048, (n + 255) / 256);
        hipLaunchKernelGGL(k_sum_peers, dim3(nblk), dim3(256), 0, h->stream, (const double* const*)c->d_peers, c->nranks, d + off, n);
        if (hipStreamSynchronize(h->stream) != hipSuccess) return fail("comm(shm): reduction kernel failed");
        if (!shm_barrier(c)) { h->last_error = "comm(shm): a peer failed or timed out after the reduction"; return FEASTHIP_ERROR_INTERNAL; }
    }
    return 0;
}

// ---- packing helpers of the per-loop reduce -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_real(const cplx* __restrict__ src, double* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i].x;
}
__global__ __launch_bounds__(256) void k_unpack_real(const double* __restrict__ src, cplx* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = cmake(src[i], 0.0);
}
void fh_launch_pack_real(const cplx* src, double* dst, size_t n, hipStream_t st) {
    const int nblk = (int)std::min<size_t>(4096, (n + 255) / 256);
    if (n) hipLaunchKernelGGL(k_pack_real, dim3(nblk), dim3(256), 0, st, src, dst, n);
}
void fh_launch_unpack_real(const double* src, cplx* dst, size_t n, hipStream_t st) {
    const int nblk = (int)std::min<size_t>(4096, (n + 255) / 256);
    if (n) hipLaunchKernelGGL(k_unpack_real, dim3(nblk), dim3(256), 0, st, src, dst, n);
}

extern "C" int feasthip_allreduce_sum_dev(feasthip_handle h, void* dptr, int64_t count) {
    if (!h || (!dptr && count > 0) || count < 0) return FEASTHIP_ERROR_INTERNAL;
    FH_CHECK(hipSetDevice(h->device));
    int rc = fh_comm_allreduce_sum(h, (double*)dptr, (size_t)count);
    if (rc) return rc;
    FH_CHECK(hipStreamSynchronize(h->stream));
    return 0;
}
