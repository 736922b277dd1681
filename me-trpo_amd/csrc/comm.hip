// Direct RCCL all-reduce from the C driver: the ~14 sum all-reduces of a TRPO iteration (loss+gradient [1+P], every
// Fisher-vector product [P], each line-search (loss, kl) pair [2]) are latency-bound float64 vectors of 8 B .. 12 KB.  With a
// communicator attached to the ctx they are issued by run_trpo_update itself on the caller's stream -- no Python, no host callback
// in the CG loop.  librccl is resolved at run time (dlopen of the copy already mapped into the process, i.e. the one PyTorch
// ships, else the system one), so libmetrpo.so itself has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <link.h>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <unistd.h>
#include <rccl/rccl.h>
#include "metrpo_internal.h"

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl g_rccl;

int find_loaded(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) { *static_cast<std::string*>(data) = info->dlpi_name; return 1; }
    return 0;
}

bool rccl_load() {
    if (g_rccl.handle) return true;
    std::string loaded;
    dl_iterate_phdr(find_loaded, &loaded);
    const char* candidates[] = {loaded.empty() ? nullptr : loaded.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : candidates) {
        if (!name) continue;
        g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) { g_rccl.err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name)); \
    if (!g_rccl.field) { g_rccl.err = std::string("librccl lacks ") + name; dlclose(g_rccl.handle); g_rccl.handle = nullptr; return false; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return true;
}
}  // namespace

extern "C" int32_t metrpo_comm_get_unique_id(void* id_out) {
    if (!id_out) return METRPO_ENULL;
    if (!rccl_load()) return METRPO_EUNSUPPORTED;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return METRPO_EHIP;
    static_assert(sizeof(ncclUniqueId) == METRPO_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id_out, &id, sizeof(id));
    return METRPO_OK;
}

extern "C" int32_t metrpo_comm_init(metrpo_ctx* c, const void* id_bytes, int32_t world, int32_t rank) {
    if (!c) return METRPO_ENULL;
    if (!id_bytes) return set_err(c, METRPO_ENULL, "comm_init: id is NULL");
    if (world < 1 || rank < 0 || rank >= world) return set_err(c, METRPO_EINVAL, "comm_init: bad world / rank");
    if (c->nccl_comm) return set_err(c, METRPO_ESTATE, "comm_init: a communicator is already attached (metrpo_comm_destroy first)");
    if (!rccl_load()) return set_err(c, METRPO_EUNSUPPORTED, g_rccl.err);
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return set_err(c, METRPO_EHIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    c->nccl_comm = comm; c->comm_world = world; c->comm_rank = rank;
    return METRPO_OK;
}

extern "C" int32_t metrpo_comm_destroy(metrpo_ctx* c) {
    if (!c) return METRPO_ENULL;
    if (c->nccl_comm) { g_rccl.CommDestroy((ncclComm_t)c->nccl_comm); c->nccl_comm = nullptr; c->comm_world = 0; c->comm_rank = 0; }
    return METRPO_OK;
}

// ---- one-shot direct all-reduce over peer-mapped receive regions (xchg_device.h) ---------------------------------------------
// Bootstrap (any side channel carries the blobs; Comm.attach_engine uses a torch.distributed all_gather):
//   every rank:  metrpo_comm_ipc_export(ctx, blob)          allocates + zeroes the local receive region, returns its IPC handle
//   all-gather the blobs
//   every rank:  metrpo_comm_ipc_attach(ctx, blobs, G, r)   maps the G-1 peer regions (hipIpcOpenMemHandle)
// The ranks may sit on different GPUs of one node (stores cross xGMI) or share one GPU (what the 1-GPU test boxes can run).
namespace {
struct IpcBlob {                       // METRPO_COMM_IPC_BLOB_BYTES
    hipIpcMemHandle_t handle;          // 64 bytes
    int32_t device, cap, magic, pid;
    char pad[METRPO_COMM_IPC_BLOB_BYTES - sizeof(hipIpcMemHandle_t) - 16];
};
static_assert(sizeof(IpcBlob) == METRPO_COMM_IPC_BLOB_BYTES, "blob size");
constexpr int32_t IPC_MAGIC = 0x58474d49;          // "XGMI"
constexpr int XG_CAP = 16384;                      // float64 elements per exchange: Humanoid's [F*F+F] = 13 110, P = 12 492

size_t region_bytes(int cap) { return (size_t)2 * XCHG_MAX_WORLD * cap * 2 * sizeof(unsigned long long); }

__global__ void k_xchg_allreduce(XchgK x, double* __restrict__ buf, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    xchg_push(x, i, buf[i]);
    buf[i] = xchg_pull_sum(x, i);
}
}  // namespace

XchgK xchg_next(metrpo_ctx* c) {
    XchgK x = {};
    if (c->xg_world <= 1) return x;
    x.world = c->xg_world; x.rank = c->xg_rank; x.cap = c->xg_cap; x.seq = ++c->xg_seq;
    if (x.seq == 0) x.seq = c->xg_seq = 2;          // wrapped after 2^32 exchanges: 0 is the stamp of untouched slots; keep the parity sequence (…, 0xffffffff (odd), 2 (even))
    for (int q = 0; q < XCHG_MAX_WORLD; ++q) x.peer[q] = (unsigned long long*)c->xg_peer[q];
    x.err = comm_err_cell(c); x.timeout_ticks = c->xg_timeout;
    return x;
}

extern "C" int32_t metrpo_comm_ipc_export(metrpo_ctx* c, void* blob_out) {
    if (!c) return METRPO_ENULL;
    if (!blob_out) return set_err(c, METRPO_ENULL, "comm_ipc_export: blob is NULL");
    if (c->xg_world > 0) return set_err(c, METRPO_ESTATE, "comm_ipc_export: a peer mapping is attached (metrpo_comm_ipc_detach first)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->xg_region) {
        // fine-grained / uncached device memory: peer writes must become visible to polling loads without a kernel boundary
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, region_bytes(XG_CAP), hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError(); p = nullptr;
            if (hipExtMallocWithFlags(&p, region_bytes(XG_CAP), hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError(); p = nullptr;
                HIP_TRY(c, hipMalloc(&p, region_bytes(XG_CAP)));
            }
        }
        c->xg_region = p; c->xg_cap = XG_CAP;
    }
    HIP_TRY(c, hipMemset(c->xg_region, 0, region_bytes(c->xg_cap)));
    HIP_TRY(c, hipMemset(comm_err_cell(c), 0, sizeof(double)));
    HIP_TRY(c, hipDeviceSynchronize());
    IpcBlob b;
    std::memset(&b, 0, sizeof(b));
    HIP_TRY(c, hipIpcGetMemHandle(&b.handle, c->xg_region));
    b.device = c->device; b.cap = c->xg_cap; b.magic = IPC_MAGIC; b.pid = (int32_t)getpid();
    std::memcpy(blob_out, &b, sizeof(b));
    return METRPO_OK;
}

extern "C" int32_t metrpo_comm_ipc_attach(metrpo_ctx* c, const void* blobs, int32_t world, int32_t rank) {
    if (!c) return METRPO_ENULL;
    if (!blobs) return set_err(c, METRPO_ENULL, "comm_ipc_attach: blobs is NULL");
    if (world < 1 || world > XCHG_MAX_WORLD || rank < 0 || rank >= world) return set_err(c, METRPO_EINVAL, "comm_ipc_attach: bad world / rank (at most 8 ranks: one node)");
    if (!c->xg_region) return set_err(c, METRPO_ESTATE, "comm_ipc_attach: call metrpo_comm_ipc_export first");
    if (c->xg_world > 0) return set_err(c, METRPO_ESTATE, "comm_ipc_attach: already attached");
    HIP_TRY(c, hipSetDevice(c->device));
    const IpcBlob* bl = static_cast<const IpcBlob*>(blobs);
    for (int q = 0; q < world; ++q)
        if (bl[q].magic != IPC_MAGIC || bl[q].cap != c->xg_cap) return set_err(c, METRPO_EINVAL, "comm_ipc_attach: malformed blob");
    for (int q = 0; q < world; ++q)                   // hipIpcOpenMemHandle refuses handles of the calling process
        if (q != rank && bl[q].pid == (int32_t)getpid()) return set_err(c, METRPO_EUNSUPPORTED, "comm_ipc_attach: two ranks in one process");
    for (int q = 0; q < XCHG_MAX_WORLD; ++q) c->xg_peer[q] = nullptr;
    for (int q = 0; q < world; ++q) {
        if (q == rank) { c->xg_peer[q] = c->xg_region; continue; }
        if (bl[q].device != c->device) {             // peer on another GPU of the node: direct access over xGMI
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, c->device, bl[q].device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(bl[q].device, 0);
                if (e != hipSuccess) (void)hipGetLastError();        // already enabled is fine; a real failure shows up in the open below
            } else (void)hipGetLastError();
        }
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, bl[q].handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j < q; ++j) if (j != rank && c->xg_peer[j]) (void)hipIpcCloseMemHandle(c->xg_peer[j]);
            for (int j = 0; j < XCHG_MAX_WORLD; ++j) c->xg_peer[j] = nullptr;
            return set_err(c, METRPO_EHIP, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(q) + "): " + hipGetErrorString(e));
        }
        c->xg_peer[q] = p;
    }
    c->xg_world = world; c->xg_rank = rank; c->xg_seq = 0; c->xg_fuse = 0;
    long long ms = 20000;
    if (const char* t = ctx_opt(c, OPT_XCHG_TIMEOUT_MS)) { const long long v = atoll(t); if (v > 0) ms = v; }
    c->xg_timeout = (unsigned long long)ms * 100000ull;       // wall_clock64: 100 MHz
    return METRPO_OK;
}

extern "C" int32_t metrpo_comm_ipc_detach(metrpo_ctx* c) {
    if (!c) return METRPO_ENULL;
    if (c->xg_world > 0) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        for (int q = 0; q < c->xg_world; ++q) if (q != c->xg_rank && c->xg_peer[q]) (void)hipIpcCloseMemHandle(c->xg_peer[q]);
    }
    for (int q = 0; q < XCHG_MAX_WORLD; ++q) c->xg_peer[q] = nullptr;
    c->xg_world = 0; c->xg_rank = 0; c->xg_fuse = 0;
    // The time-out cell of the exchanges is sticky by design; it must not outlive the transport that raised it: 'auto' mode detaches after a
    // timed-out test exchange and goes on over RCCL or the caller's callback, and every later update of this context would report that old time-out.
    if (c->d_cg) {
        (void)hipSetDevice(c->device);
        (void)hipMemset(comm_err_cell(c), 0, sizeof(double));
        (void)hipDeviceSynchronize();
    }
    return METRPO_OK;
}

extern "C" int32_t metrpo_comm_set_timeout_ms(metrpo_ctx* c, int64_t ms) {
    if (!c) return METRPO_ENULL;
    if (ms <= 0) return set_err(c, METRPO_EINVAL, "comm_set_timeout_ms: must be positive");
    c->xg_timeout = (unsigned long long)ms * 100000ull;
    return METRPO_OK;
}

// 0 = single rank, 1 = RCCL communicator, 2 = one-shot direct all-reduce over peer-mapped regions
extern "C" int32_t metrpo_comm_transport(const metrpo_ctx* c) {
    if (!c) return METRPO_ENULL;
    return c->xg_world > 1 ? 2 : (c->nccl_comm ? 1 : 0);
}

// synchronises `stream` and reports whether any exchange so far ran into its time limit (a rank that never arrived)
extern "C" int32_t metrpo_comm_check(metrpo_ctx* c, void* stream) {
    if (!c) return METRPO_ENULL;
    HIP_TRY(c, hipMemcpyAsync(c->h_pinned + 14, comm_err_cell(c), 2 * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)stream));
    if (c->h_pinned[15] != 0.0) return rollout_error_seen(c, (hipStream_t)stream);
    if (c->h_pinned[14] != 0.0) return set_err(c, METRPO_EHIP, "one-shot all-reduce: a rank did not arrive within the time limit (METRPO_XCHG_TIMEOUT_MS)");
    return METRPO_OK;
}

int comm_allreduce_f64(metrpo_ctx* c, double* buf, long long count, hipStream_t st) {
    if (c->xg_world > 1) {
        for (long long off = 0; off < count; off += c->xg_cap) {           // vectors longer than a slot go in slot-sized pieces
            const int n = (int)std::min<long long>(c->xg_cap, count - off);
            hipLaunchKernelGGL(k_xchg_allreduce, dim3((n + 255) / 256), dim3(256), 0, st, xchg_next(c), buf + off, n);
        }
        HIP_TRY(c, hipGetLastError());
        return METRPO_OK;
    }
    if (!c->nccl_comm) return set_err(c, METRPO_ESTATE, "all-reduce: no communicator attached (metrpo_comm_init)");
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)c->nccl_comm, st);
    if (r != ncclSuccess) return set_err(c, METRPO_EHIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    return METRPO_OK;
}

extern "C" int32_t metrpo_allreduce_sum_f64(metrpo_ctx* c, double* d_buf, int64_t count, void* stream) {
    if (!c) return METRPO_ENULL;
    if (!d_buf) return set_err(c, METRPO_ENULL, "allreduce: NULL buffer");
    if (count < 0) return set_err(c, METRPO_EINVAL, "allreduce: negative count");
    if (count == 0) return METRPO_OK;
    return comm_allreduce_f64(c, d_buf, count, (hipStream_t)stream);
}
