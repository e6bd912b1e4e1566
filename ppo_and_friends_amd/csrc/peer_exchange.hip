// K17: per-mini-batch gradient exchange between the ranks of one node over peer mappings (xGMI).
//
// Replaces, inside the update loop, the Allreduce of mpi_avg_gradients (utils/mpi_utils.py:65-86 of the
// reference, called from ppo.py:2443-2448 once per mini-batch): the bucket is ~271 KB at C2, so the exchange is
// pure latency.  One launch per exchange, no library call and no host round trip -- the launch has fixed
// arguments and can be captured in a hipGraph together with the update kernels:
//
//   phase 1  workgroup g copies its elements of the local gradient into this rank's exchange slot (uncached
//            device memory exported to the peers by IPC; two slots, alternating by sequence number), fences
//            them to system scope and stores its sequence number into every peer's flag word for
//            (group g, this rank) -- posted remote stores,
//   phase 2  it polls its LOCAL flag words until group g of every peer has published this sequence number,
//   phase 3  it reads its elements from all the peers' slots (loads of up to 8 peers in flight) and adds them in
//            rank order -- every rank computes the bitwise identical sum, so replicas stay identical -- writes
//            the summed bucket and its squared-norm partial; the partials are folded in a fixed association
//            (deterministic clip coefficient on every rank) by the last workgroup to finish, or by the
//            consuming kernel (K12's Adam launch).
//
// Group g of every rank owns the same elements, so a workgroup depends only on its peers' group g: there is no
// rank-wide arrival counter (each would cost a ~1.5 us memory-side atomic round trip) and no single publisher.
// Two slots suffice with one flag per (group, exchange): a group rewrites its part of slot s&1 at exchange s+2
// only after it has passed its wait of exchange s+1, and the peer's group publishes s+1 only after its
// exchange-s launch -- hence its reads of that part -- has completed.
// Every wait is bounded (wall-clock budget): on expiry the error word is set and the launch drains.
#include "peer_exchange_device.hpp"

#include <unistd.h>
#include <cstring>
#include <new>

namespace ppoaf {

__global__ __launch_bounds__(kXchgThreads) void peer_allreduce_kernel(XchgDev x, const float4* src, float4* dst,
                                                                     long split4, float norm_scale,
                                                                     double* norm_out, long long wait_ticks) {
    __shared__ double red[17];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const unsigned g = blockIdx.x;                   // group g owns elements g*256 + tid + k*grid*256, on every rank
    const long long seq = xchg_sequence(x, g);
    const long slot = (long)(seq & 1) * x.n4;
    const long stride = (long)gridDim.x * kXchgThreads;
    float4* mine = x.my_slots + slot;
    for (long i = (long)g * kXchgThreads + tid; i < x.n4; i += stride) mine[i] = src[i];
    xchg_publish(x, seq, g);
    xchg_wait(x, seq, g, wait_ticks);
    double q0 = 0.0, q1 = 0.0;
    for (long i = (long)g * kXchgThreads + tid; i < x.n4; i += stride) {
        const float4 acc = xchg_sum(x, slot, i, src[i]);
        dst[i] = acc;
        const double q = xchg_sq(acc, norm_scale);
        if (i < split4) q0 += q; else q1 += q;
    }
    xchg_advance(x, seq, g);
    if (norm_out) {                                  // uniform over the launch
        q0 = block_sum(q0, red);
        q1 = block_sum(q1, red);
        xchg_finish(x, g, gridDim.x, q0, q1, split4 < x.n4, norm_out, &s_last);
    }
}

}  // namespace ppoaf

namespace ppoaf {

struct ExportBlob {                                 // PPOAF_PEER_EXCHANGE_BLOB_BYTES
    hipIpcMemHandle_t handle;                       // 64 bytes
    int64_t n4;
    int32_t rank, n_ranks;
    int64_t pid;
    char pad[PPOAF_PEER_EXCHANGE_BLOB_BYTES - 64 - 8 - 8 - 8];
};
static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
static_assert(sizeof(ExportBlob) == PPOAF_PEER_EXCHANGE_BLOB_BYTES, "blob layout");

}  // namespace ppoaf

using namespace ppoaf;

#define PPOAF_HIP_TRY(expr, what)                                                      \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s: %s", what, hipGetErrorString(e_));                          \
            return PPOAF_E_LAUNCH;                                                     \
        }                                                                              \
    } while (0)

extern "C" int ppoaf_peer_exchange_create(int rank, int n_ranks, int64_t bucket_floats, int memory_kind,
                                          ppoaf_peer_exchange_t** out) {
    PPOAF_REQUIRE(out, "peer_exchange_create: null out");
    PPOAF_REQUIRE(n_ranks >= 1 && n_ranks <= kMaxPeers && rank >= 0 && rank < n_ranks,
                  "peer_exchange_create: rank %d of %d (at most %d ranks)", rank, n_ranks, kMaxPeers);
    PPOAF_REQUIRE(bucket_floats > 0 && bucket_floats % 4 == 0, "peer_exchange_create: bucket of %ld floats (multiple of 4)",
                  (long)bucket_floats);
    PPOAF_REQUIRE(memory_kind >= 0 && memory_kind <= 3, "peer_exchange_create: memory_kind=%d (0 auto, 1 uncached, 2 fine-grained, 3 coarse)",
                  memory_kind);
    ppoaf_peer_exchange* x = new (std::nothrow) ppoaf_peer_exchange();
    PPOAF_REQUIRE(x, "peer_exchange_create: out of host memory");
    x->dev.rank = rank; x->dev.n_ranks = n_ranks; x->dev.n4 = bucket_floats / 4;
    x->bytes = kXchgHeaderBytes + 2 * (size_t)bucket_floats * sizeof(float);
    // exchange memory: 1 uncached (default: remote reads / writes never meet a cached copy), 2 fine-grained,
    // 3 ordinary coarse-grained device memory (relies on the system-scope fences of the kernel alone);
    // 0 = uncached, or fine-grained if that allocation is refused
    hipError_t e = hipErrorInvalidValue;
    if (memory_kind == 0 || memory_kind == 1) {
        e = hipExtMallocWithFlags(&x->base, x->bytes, hipDeviceMallocUncached);
        x->memory_kind = 1;
    }
    if ((memory_kind == 0 && e != hipSuccess) || memory_kind == 2) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&x->base, x->bytes, hipDeviceMallocFinegrained);
        x->memory_kind = 2;
    }
    if (memory_kind == 3) {
        e = hipMalloc(&x->base, x->bytes);
        x->memory_kind = 3;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("peer_exchange_create: exchange memory (kind %d): %s", memory_kind, hipGetErrorString(e));
        delete x;
        return PPOAF_E_LAUNCH;
    }
    void* local = nullptr;
    const size_t local_bytes = (4 + kXchgMaxGroups) * sizeof(long long) + 2 * kXchgMaxGroups * sizeof(double);
    e = hipMalloc(&local, local_bytes);
    if (e == hipSuccess) e = hipMemset(local, 0, local_bytes);
    if (e == hipSuccess) e = hipMemset(x->base, 0, x->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_error("peer_exchange_create: %s", hipGetErrorString(e));
        if (local) (void)hipFree(local);
        (void)hipFree(x->base);
        delete x;
        return PPOAF_E_LAUNCH;
    }
    x->dev.words = static_cast<long long*>(local);
    x->dev.group_seq = x->dev.words + 4;
    x->dev.norm_partials = reinterpret_cast<double*>(x->dev.group_seq + kXchgMaxGroups);
    x->dev.my_flags = static_cast<long long*>(x->base);
    x->dev.my_slots = reinterpret_cast<float4*>(static_cast<char*>(x->base) + kXchgHeaderBytes);
    for (int p = 0; p < kMaxPeers; ++p) { x->dev.peer_flags[p] = nullptr; x->dev.peer_slots[p] = nullptr; }
    x->dev.peer_flags[rank] = x->dev.my_flags;
    x->dev.peer_slots[rank] = x->dev.my_slots;
    x->connected = n_ranks == 1;
    *out = x;
    return PPOAF_OK;
}

extern "C" int ppoaf_peer_exchange_export(ppoaf_peer_exchange_t* x, void* blob) {
    PPOAF_REQUIRE(x && blob, "peer_exchange_export: null argument");
    ExportBlob b;
    memset(&b, 0, sizeof(b));
    PPOAF_HIP_TRY(hipIpcGetMemHandle(&b.handle, x->base), "peer_exchange_export: hipIpcGetMemHandle");
    b.n4 = x->dev.n4; b.rank = x->dev.rank; b.n_ranks = x->dev.n_ranks; b.pid = (int64_t)getpid();
    memcpy(blob, &b, sizeof(b));
    return PPOAF_OK;
}

extern "C" int ppoaf_peer_exchange_connect(ppoaf_peer_exchange_t* x, const void* all_blobs) {
    PPOAF_REQUIRE(x && all_blobs, "peer_exchange_connect: null argument");
    PPOAF_REQUIRE(!x->connected || x->dev.n_ranks == 1, "peer_exchange_connect: already connected");
    const ExportBlob* blobs = static_cast<const ExportBlob*>(all_blobs);
    for (int p = 0; p < x->dev.n_ranks; ++p) {
        ExportBlob b;
        memcpy(&b, &blobs[p], sizeof(b));
        PPOAF_REQUIRE(b.rank == p && b.n_ranks == x->dev.n_ranks && b.n4 == x->dev.n4,
                      "peer_exchange_connect: blob %d describes rank %d of %d with %ld float4 (expected %d ranks, %ld)",
                      p, b.rank, b.n_ranks, (long)b.n4, x->dev.n_ranks, x->dev.n4);
        if (p == x->dev.rank) continue;
        PPOAF_REQUIRE(b.pid != (int64_t)getpid(), "peer_exchange_connect: rank %d lives in this process", p);
        void* ptr = nullptr;
        PPOAF_HIP_TRY(hipIpcOpenMemHandle(&ptr, b.handle, hipIpcMemLazyEnablePeerAccess),
                      "peer_exchange_connect: hipIpcOpenMemHandle");
        x->opened[p] = ptr;
        x->dev.peer_flags[p] = static_cast<long long*>(ptr);
        x->dev.peer_slots[p] = reinterpret_cast<const float4*>(static_cast<char*>(ptr) + kXchgHeaderBytes);
    }
    x->connected = true;
    return PPOAF_OK;
}

extern "C" int ppoaf_peer_exchange_allreduce(ppoaf_peer_exchange_t* x, const float* src, float* dst,
                                             int64_t split_floats, float norm_scale, double* norm_out,
                                             double wait_seconds, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(x && src && dst, "peer_exchange_allreduce: null argument");
    PPOAF_REQUIRE(x->connected, "peer_exchange_allreduce: not connected");
    PPOAF_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "peer_exchange_allreduce: 16-byte alignment");
    PPOAF_REQUIRE(split_floats >= 0 && split_floats % 4 == 0, "peer_exchange_allreduce: split=%ld (multiple of 4)",
                  (long)split_floats);
    PPOAF_REQUIRE(wait_seconds > 0.0 && wait_seconds <= 600.0, "peer_exchange_allreduce: wait_seconds=%g", wait_seconds);
    long blocks = (x->dev.n4 + kXchgThreads - 1) / kXchgThreads;
    if (blocks > kXchgMaxGrid) blocks = kXchgMaxGrid;    // one flag word per group; every launch of this object uses this grid
    const long long ticks = (long long)(wait_seconds * 1.0e8);     // wall_clock64: 100 MHz
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3((unsigned)blocks), dim3(kXchgThreads), 0, (hipStream_t)stream,
                       x->dev, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst),
                       (long)(split_floats / 4), norm_scale, norm_out, ticks);
    return check_launch("peer_exchange_allreduce");
}

extern "C" int ppoaf_peer_exchange_status(ppoaf_peer_exchange_t* x, int64_t* out) {
    PPOAF_REQUIRE(x && out, "peer_exchange_status: null argument");
    long long w[4];
    PPOAF_HIP_TRY(hipMemcpy(w, x->dev.words, sizeof(w), hipMemcpyDeviceToHost), "peer_exchange_status: hipMemcpy");
    out[0] = w[0]; out[1] = w[3]; out[2] = x->memory_kind; out[3] = x->dev.n_ranks;
    return PPOAF_OK;
}

extern "C" int ppoaf_peer_exchange_destroy(ppoaf_peer_exchange_t* x) {
    if (!x) return PPOAF_OK;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < kMaxPeers; ++p)
        if (x->opened[p]) (void)hipIpcCloseMemHandle(x->opened[p]);
    if (x->dev.words) (void)hipFree(x->dev.words);
    if (x->base) (void)hipFree(x->base);
    delete x;
    return PPOAF_OK;
}
