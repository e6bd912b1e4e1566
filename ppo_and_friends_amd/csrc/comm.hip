// Rank collectives behind the C ABI (ppoaf_comm_init / allreduce_avg / bcast / moments).
//
// Replaces, for a host that binds this library directly instead of going through torch.distributed,
//   broadcast_model_parameters   utils/mpi_utils.py:50-63   (per-tensor comm.Bcast)      -> ppoaf_bcast_f32 on the flat bucket
//   mpi_avg / mpi_avg_gradients  utils/mpi_utils.py:65-111  (per-tensor comm.Allreduce)  -> ppoaf_allreduce_avg_f32
//   RunningMeanStd.update        utils/stats.py:47-50       (comm.allgather of raw data) -> ppoaf_allgather_moments of
//                                                                                          the (n, mean, M2) records
// over RCCL (one communicator per process, one process per GPU; xGMI inside a node).  RCCL is bound at
// run time -- dlopen of the librccl already mapped by the process (torch-ROCm ships one) or the ROCm one --
// so that the library has no link-time dependency on it and every other entry point works without RCCL.
// The per-mini-batch gradient exchange of the update loop has its own, lower-latency path (K17,
// peer_exchange.hip); these collectives are the general ones (any node count).
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <new>

namespace ppoaf {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }      // the copy already mapped
    if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return nullptr;
#define PPOAF_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name))
    PPOAF_SYM(GetUniqueId, "ncclGetUniqueId"); PPOAF_SYM(CommInitRank, "ncclCommInitRank");
    PPOAF_SYM(CommDestroy, "ncclCommDestroy"); PPOAF_SYM(AllReduce, "ncclAllReduce");
    PPOAF_SYM(Broadcast, "ncclBroadcast"); PPOAF_SYM(AllGather, "ncclAllGather");
    PPOAF_SYM(GetErrorString, "ncclGetErrorString");
#undef PPOAF_SYM
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.Broadcast || !api.AllGather)
        return nullptr;
    api.handle = h;
    return &api;
}

__global__ void scale_f32_kernel(float* __restrict__ x, long n, float s) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= s;
}

}  // namespace ppoaf

using namespace ppoaf;

struct ppoaf_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

#define PPOAF_RCCL_TRY(expr, what)                                                                       \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) {                                                                         \
            set_error("%s: %s", what, api->GetErrorString ? api->GetErrorString(r_) : "RCCL error");     \
            return PPOAF_E_LAUNCH;                                                                       \
        }                                                                                                \
    } while (0)

extern "C" int ppoaf_comm_unique_id(void* out) {
    PPOAF_REQUIRE(out, "comm_unique_id: null out");
    RcclApi* api = rccl_api();
    PPOAF_REQUIRE(api, "comm_unique_id: RCCL (librccl.so) is not available in this process");
    ncclUniqueId id;
    PPOAF_RCCL_TRY(api->GetUniqueId(&id), "comm_unique_id");
    static_assert(sizeof(id) == PPOAF_COMM_UNIQUE_ID_BYTES, "unique id size");
    memcpy(out, &id, sizeof(id));
    return PPOAF_OK;
}

extern "C" int ppoaf_comm_init(int rank, int world, const void* unique_id, ppoaf_comm_t** out) {
    PPOAF_REQUIRE(out && unique_id, "comm_init: null argument");
    PPOAF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    RcclApi* api = rccl_api();
    PPOAF_REQUIRE(api, "comm_init: RCCL (librccl.so) is not available in this process");
    ppoaf_comm* c = new (std::nothrow) ppoaf_comm();
    PPOAF_REQUIRE(c, "comm_init: out of host memory");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error("comm_init: %s", api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
        delete c;
        return PPOAF_E_LAUNCH;
    }
    c->rank = rank; c->world = world;
    *out = c;
    return PPOAF_OK;
}

extern "C" int ppoaf_allreduce_avg_f32(ppoaf_comm_t* c, float* buf, int64_t n, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(c && c->comm, "allreduce_avg_f32: communicator missing");
    PPOAF_REQUIRE(n >= 0, "allreduce_avg_f32: n=%ld", (long)n);
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(buf, "allreduce_avg_f32: null buffer");
    RcclApi* api = rccl_api();
    hipStream_t s = (hipStream_t)stream;
    PPOAF_RCCL_TRY(api->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, s), "allreduce_avg_f32");
    if (c->world > 1) {                          // the reference divides by num_procs after the sum (mpi_utils.py:86)
        long blocks = (n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(scale_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, buf, (long)n, 1.0f / (float)c->world);
        return check_launch("allreduce_avg_f32/scale");
    }
    return PPOAF_OK;
}

extern "C" int ppoaf_allreduce_sum_f32(ppoaf_comm_t* c, float* buf, int64_t n, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(c && c->comm, "allreduce_sum_f32: communicator missing");
    PPOAF_REQUIRE(n >= 0, "allreduce_sum_f32: n=%ld", (long)n);
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(buf, "allreduce_sum_f32: null buffer");
    RcclApi* api = rccl_api();
    PPOAF_RCCL_TRY(api->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream), "allreduce_sum_f32");
    return PPOAF_OK;
}

extern "C" int ppoaf_bcast_f32(ppoaf_comm_t* c, float* buf, int64_t n, int root, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(c && c->comm, "bcast_f32: communicator missing");
    PPOAF_REQUIRE(n >= 0 && root >= 0 && root < c->world, "bcast_f32: n=%ld root=%d", (long)n, root);
    if (n == 0) return PPOAF_OK;
    PPOAF_REQUIRE(buf, "bcast_f32: null buffer");
    RcclApi* api = rccl_api();
    PPOAF_RCCL_TRY(api->Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->comm, (hipStream_t)stream), "bcast_f32");
    return PPOAF_OK;
}

extern "C" int ppoaf_allgather_moments(ppoaf_comm_t* c, const double* record, int64_t n_doubles, double* out,
                                       ppoaf_stream_t stream) {
    PPOAF_REQUIRE(c && c->comm, "allgather_moments: communicator missing");
    PPOAF_REQUIRE(record && out && n_doubles >= 1, "allgather_moments: null argument or n_doubles=%ld", (long)n_doubles);
    RcclApi* api = rccl_api();
    PPOAF_RCCL_TRY(api->AllGather(record, out, (size_t)n_doubles, ncclFloat64, c->comm, (hipStream_t)stream),
                   "allgather_moments");
    return PPOAF_OK;
}

extern "C" int ppoaf_comm_destroy(ppoaf_comm_t* c) {
    if (!c) return PPOAF_OK;
    RcclApi* api = rccl_api();
    if (api && c->comm) (void)api->CommDestroy(c->comm);
    delete c;
    return PPOAF_OK;
}
