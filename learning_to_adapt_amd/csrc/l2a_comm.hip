// l2a_comm.hip - the one collective of a sharded plan step in the C ABI (include/l2a.h): an in-place MAX all-reduce
// of the packed arg-max keys over RCCL (xGMI inside a node).  Host code only.
//
// RCCL is bound at run time (dlopen of librccl.so on the first l2a_comm_* call): libl2a_hip.so has no link-time
// dependency on it, so a single-GPU host - or a box without RCCL - loads the library and plans as before.

#include "l2a_host.h"

#include <dlfcn.h>

#include <cstring>
#include <string>

// ---- payload of a sharded plan's collective (device side: one wave) -----------------------------------------------
__global__ void l2a_plan_payload_k(const unsigned long long* best_key, int m, const unsigned int* status,
                                   unsigned long long digest, unsigned long long* payload) {
    for (int i = threadIdx.x; i < m; i += blockDim.x) payload[i] = best_key[i];
    if (threadIdx.x == 0) {
        // the status word lives in host-mapped memory; the rollout kernel ORed into it with system scope and has
        // completed (stream order)
        const unsigned int st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long d = digest & L2A_DIGEST_MASK;
        payload[m] = st ? 1ull : 0ull;
        payload[m + 1] = d;
        payload[m + 2] = L2A_DIGEST_MASK - d;
    }
}

namespace {

// the subset of rccl.h used here (ABI-stable: NCCL 2.x)
typedef struct l2a_nccl_comm* nccl_comm_t;
typedef struct { char internal[128]; } nccl_unique_id;       // NCCL_UNIQUE_ID_BYTES
enum { NCCL_SUCCESS = 0, NCCL_MAX = 2, NCCL_UINT64 = 5 };    // ncclResult_t / ncclRedOp_t / ncclDataType_t values

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    if (r.handle || !r.error.empty()) return &r;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) { r.error = std::string("librccl.so not found: ") + dlerror(); return &r; }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString)
        r.error = "librccl.so lacks an expected symbol";
    return &r;
}

int nccl_fail(l2a_ctx* ctx, const char* what, int rc) {
    Rccl* r = rccl();
    return l2a_fail(ctx, L2A_EHIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
}

}  // namespace

extern "C" {

int l2a_comm_unique_id(char* id_out) {
    if (!id_out) return L2A_EINVAL;
    Rccl* r = rccl();
    if (!r->error.empty()) return l2a_fail(nullptr, L2A_ENODEV, r->error);
    nccl_unique_id id;
    const int rc = r->GetUniqueId(&id);
    if (rc != NCCL_SUCCESS) return nccl_fail(nullptr, "ncclGetUniqueId", rc);
    std::memcpy(id_out, id.internal, sizeof(id.internal));
    return L2A_OK;
}

int l2a_comm_init(l2a_ctx* ctx, int rank, int world, const char* id_bytes) {
    if (!ctx) return L2A_EINVAL;
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) return l2a_fail(ctx, L2A_EINVAL, "l2a_comm_init: bad rank / world / id");
    if (ctx->comm) return l2a_fail(ctx, L2A_ESTATE, "l2a_comm_init: this context already has a communicator");
    Rccl* r = rccl();
    if (!r->error.empty()) return l2a_fail(ctx, L2A_ENODEV, r->error);
    l2a_device_guard guard(ctx->device);
    L2A_HIP(ctx, hipSetDevice(ctx->device));        // ncclCommInitRank binds the communicator to the current device
    nccl_unique_id id;
    std::memcpy(id.internal, id_bytes, sizeof(id.internal));
    nccl_comm_t comm = nullptr;
    const int rc = r->CommInitRank(&comm, world, id, rank);
    if (rc != NCCL_SUCCESS) return nccl_fail(ctx, "ncclCommInitRank", rc);
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return L2A_OK;
}

int l2a_comm_destroy(l2a_ctx* ctx) {
    if (!ctx) return L2A_EINVAL;
    if (!ctx->comm) return L2A_OK;
    Rccl* r = rccl();
    const int rc = r->CommDestroy(static_cast<nccl_comm_t>(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    if (rc != NCCL_SUCCESS) return nccl_fail(ctx, "ncclCommDestroy", rc);
    return L2A_OK;
}

int l2a_allreduce_best(l2a_ctx* ctx, unsigned long long* best_key, int m, void* stream_v) {
    if (!ctx) return L2A_EINVAL;
    if (!best_key || m < 1) return l2a_fail(ctx, L2A_EINVAL, "l2a_allreduce_best: null keys or m < 1");
    if (!ctx->comm) return l2a_fail(ctx, L2A_ESTATE, "l2a_allreduce_best: call l2a_comm_init first");
    Rccl* r = rccl();
    l2a_device_guard guard(ctx->device);
    // keys are (orderable_u32(return) << 31) | (0x7fffffff - global_index) with the top bit clear: unsigned max ==
    // "largest return, ties -> lowest global index" == np.argmax over the concatenated shards (mpc_controller.py:129)
    const int rc = r->AllReduce(best_key, best_key, (size_t)m, NCCL_UINT64, NCCL_MAX, static_cast<nccl_comm_t>(ctx->comm),
                                reinterpret_cast<hipStream_t>(stream_v));
    if (rc != NCCL_SUCCESS) return nccl_fail(ctx, "ncclAllReduce", rc);
    return L2A_OK;
}

}  // extern "C"

extern "C" int l2a_plan_payload(l2a_ctx* ctx, const unsigned long long* best_key, int m, unsigned long long digest,
                                unsigned long long* payload, void* stream) {
    if (!ctx) return L2A_EINVAL;
    if (!best_key || !payload || m < 1) return l2a_fail(ctx, L2A_EINVAL, "l2a_plan_payload: null pointer or m < 1");
    l2a_device_guard guard(ctx->device);
    hipLaunchKernelGGL(l2a_plan_payload_k, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), best_key, m,
                       ctx->status_dev, digest, payload);
    L2A_HIP(ctx, hipGetLastError());
    return L2A_OK;
}
