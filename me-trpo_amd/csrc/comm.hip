// Direct RCCL all-reduce from the C driver: the ~14 sum all-reduces of a TRPO iteration (loss+gradient [1+P], every
// Fisher-vector product [P], each line-search (loss, kl) pair [2]) are latency-bound float64 vectors of 8 B .. 12 KB.  With a
// communicator attached to the ctx they are issued by run_trpo_update itself on the caller's stream -- no Python, no host callback
// in the CG loop.  librccl is resolved at run time (dlopen of the copy already mapped into the process, i.e. the one PyTorch
// ships, else the system one), so libmetrpo.so itself has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <link.h>
#include <cstring>
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

int comm_allreduce_f64(metrpo_ctx* c, double* buf, long long count, hipStream_t st) {
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
