// The single exchange step of the multi-GPU path: an in-place fp64 sum
// all-reduce of the accumulator record over RCCL (xGMI inside one node).
// librccl is opened lazily so that single-GPU use never touches it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "vs_common.h"

namespace {
struct rccl_api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
rccl_api g_rccl;

int load_rccl() {
    if (g_rccl.handle) return VS_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return vs_fail(VS_ERR_COMM, "cannot open librccl: %s", dlerror());
#define VS_SYM(field, name)                                                    \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));   \
    if (!g_rccl.field) return vs_fail(VS_ERR_COMM, "librccl lacks %s", name)
    VS_SYM(GetUniqueId, "ncclGetUniqueId");
    VS_SYM(CommInitRank, "ncclCommInitRank");
    VS_SYM(AllReduce, "ncclAllReduce");
    VS_SYM(CommDestroy, "ncclCommDestroy");
    VS_SYM(GetErrorString, "ncclGetErrorString");
#undef VS_SYM
    g_rccl.handle = h;
    return VS_OK;
}
}  // namespace

#define VS_NCCL(call)                                                                         \
    do {                                                                                      \
        ncclResult_t r__ = (call);                                                            \
        if (r__ != ncclSuccess)                                                               \
            return vs_fail(VS_ERR_COMM, "%s failed: %s", #call, g_rccl.GetErrorString(r__));  \
    } while (0)

static_assert(VS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

extern "C" int vs_comm_unique_id(void* id_out) {
    VS_GUARD_BEGIN
    VS_REQUIRE(id_out, "id_out is null");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    VS_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_comm_init(vs_context* ctx, const void* id, int nranks, int rank) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && id, "null argument");
    VS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d not in [0,%d)", rank, nranks);
    if (ctx->comm) return vs_fail(VS_ERR_STATE, "communicator already initialised");
    int rc = load_rccl();
    if (rc) return rc;
    VS_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    VS_NCCL(g_rccl.CommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm;
    ctx->comm_ranks = nranks;
    ctx->comm_rank = rank;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_comm_allreduce(vs_context* ctx, double* buf_dev, size_t count) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && buf_dev, "null argument");
    if (!ctx->comm) return vs_fail(VS_ERR_STATE, "vs_comm_init has not been called on this context");
    vs_launch_scope scope(ctx, VS_KERNEL_ALLREDUCE);
    VS_NCCL(g_rccl.AllReduce(buf_dev, buf_dev, count, ncclFloat64, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_comm_destroy(vs_context* ctx) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx, "ctx is null");
    if (!ctx->comm) return VS_OK;
    hipStreamSynchronize(ctx->stream);
    ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_ranks = 0;
    ctx->comm_rank = -1;
    if (r != ncclSuccess) return vs_fail(VS_ERR_COMM, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
    return VS_OK;
    VS_GUARD_END
}
