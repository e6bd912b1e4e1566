// K17: per-mini-batch gradient exchange between the ranks of one node over peer mappings (xGMI).
//
// Replaces, inside the update loop, the Allreduce of mpi_avg_gradients (utils/mpi_utils.py:65-86 of the
// reference, called from ppo.py:2443-2448 once per mini-batch): the bucket is ~136 KB, so the exchange is
// pure latency.  One launch per exchange, no library call and no host round trip -- the launch has fixed
// arguments and can be captured in a hipGraph together with the update kernels:
//
//   phase 1  every workgroup copies its slice of the local gradient into this rank's exchange slot
//            (uncached device memory exported to the peers by IPC; two slots, alternating by sequence
//            number) and fences it to system scope; the last workgroup to arrive stores the sequence
//            number into every peer's flag word (posted remote stores),
//   phase 2  the workgroups poll the LOCAL flag words until every peer has published this sequence number,
//   phase 3  every workgroup reads its slice from all the peers' slots (loads of up to 8 peers in flight) and
//            adds them in rank order -- every rank computes the bitwise identical sum, so replicas stay
//            identical -- writes the summed bucket and per-workgroup squared-norm partials; the last
//            workgroup to finish adds the partials in workgroup order (deterministic clip coefficient on
//            every rank) and advances the sequence word.
//
// Two slots suffice with one flag per exchange: a rank rewrites slot s&1 at exchange s+2 only after it has
// passed the wait of exchange s+1, and a peer publishes s+1 only after its exchange-s launch has completed.
// Every wait is bounded (wall-clock budget): on expiry the error word is set and the launch drains.
#include "common.hpp"

#include <unistd.h>
#include <cstring>
#include <new>

namespace ppoaf {

constexpr int kMaxPeers = PPOAF_PEER_EXCHANGE_MAX_RANKS;
constexpr int kXchgThreads = 256;
constexpr int kXchgMaxGrid = 64;
constexpr size_t kHeaderBytes = 256;                // flag words of one rank; slots start after it

struct XchgDev {
    int rank, n_ranks;
    long n4;                                        // float4 elements of one slot
    long long* words;                               // local: [0] sequence, [1] arrive count, [2] finish count, [3] error
    long long* my_flags;                            // this rank's flag words (polled locally)
    long long* peer_flags[kMaxPeers];               // rank p's flag words (remote store target)
    const float4* peer_slots[kMaxPeers];            // rank p's two slots
    float4* my_slots;
    double* norm_partials;                          // [kXchgMaxGrid][2]
};

__global__ __launch_bounds__(kXchgThreads) void peer_allreduce_kernel(XchgDev x, const float4* src, float4* dst,
                                                                     long split4, float norm_scale,
                                                                     double* norm_out, long long wait_ticks) {
    __shared__ double red[17];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const long long seq = x.words[0] + 1;           // advanced by the last workgroup of the previous launch
    const long slot = (long)(seq & 1) * x.n4;
    const long stride = (long)gridDim.x * kXchgThreads;
    // ---- phase 1: publish
    float4* mine = x.my_slots + slot;
    for (long i = (long)blockIdx.x * kXchgThreads + tid; i < x.n4; i += stride) mine[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned long long prev = atomicAdd(reinterpret_cast<unsigned long long*>(&x.words[1]), 1ull);
        if (prev == (unsigned long long)gridDim.x - 1ull) {
            __hip_atomic_store(&x.words[1], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int p = 0; p < x.n_ranks; ++p)
                if (p != x.rank)
                    __hip_atomic_store(&x.peer_flags[p][x.rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- phase 2: wait for every peer's sequence number
    if (tid < x.n_ranks && tid != x.rank) {
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(&x.my_flags[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(2);
            if ((long long)wall_clock64() - t0 > wait_ticks) {
                __hip_atomic_store(&x.words[3], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);        // system scope: drop any cached peer lines
    // ---- phase 3: sum in rank order
    double q0 = 0.0, q1 = 0.0;
    for (long i = (long)blockIdx.x * kXchgThreads + tid; i < x.n4; i += stride) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p0 = 0; p0 < x.n_ranks; p0 += 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int p = p0 + k;
                if (p >= x.n_ranks) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (p == x.rank) v[k] = src[i];
                else v[k] = x.peer_slots[p][slot + i];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (p0 + k < x.n_ranks) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
        dst[i] = acc;
        const float sc = norm_scale;
        const double q = (double)(acc.x * sc) * (acc.x * sc) + (double)(acc.y * sc) * (acc.y * sc) +
                         (double)(acc.z * sc) * (acc.z * sc) + (double)(acc.w * sc) * (acc.w * sc);
        if (i < split4) q0 += q; else q1 += q;
    }
    q0 = block_sum(q0, red);
    q1 = block_sum(q1, red);
    if (tid == 0) {
        x.norm_partials[2 * blockIdx.x] = q0;
        x.norm_partials[2 * blockIdx.x + 1] = q1;
        __threadfence();
        const unsigned long long prev = atomicAdd(reinterpret_cast<unsigned long long*>(&x.words[2]), 1ull);
        s_last = prev == (unsigned long long)gridDim.x - 1ull;
    }
    __syncthreads();
    if (s_last && tid == 0) {
        __threadfence();
        double n0 = 0.0, n1 = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b) {
            n0 += __hip_atomic_load(&x.norm_partials[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            n1 += __hip_atomic_load(&x.norm_partials[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (norm_out) { norm_out[0] = n0; if (split4 < x.n4) norm_out[1] = n1; }     // one segment: one word
        __hip_atomic_store(&x.words[2], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&x.words[0], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

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

struct ppoaf_peer_exchange {
    XchgDev dev;
    void* base = nullptr;                           // exported allocation: header + 2 slots
    size_t bytes = 0;
    void* opened[kMaxPeers] = {};
    bool connected = false;
    int memory_kind = 0;                            // 1 uncached, 2 fine-grained
};

#define PPOAF_HIP_TRY(expr, what)                                                      \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s: %s", what, hipGetErrorString(e_));                          \
            return PPOAF_E_LAUNCH;                                                     \
        }                                                                              \
    } while (0)

extern "C" int ppoaf_peer_exchange_create(int rank, int n_ranks, int64_t bucket_floats,
                                          ppoaf_peer_exchange_t** out) {
    PPOAF_REQUIRE(out, "peer_exchange_create: null out");
    PPOAF_REQUIRE(n_ranks >= 1 && n_ranks <= kMaxPeers && rank >= 0 && rank < n_ranks,
                  "peer_exchange_create: rank %d of %d (at most %d ranks)", rank, n_ranks, kMaxPeers);
    PPOAF_REQUIRE(bucket_floats > 0 && bucket_floats % 4 == 0, "peer_exchange_create: bucket of %ld floats (multiple of 4)",
                  (long)bucket_floats);
    ppoaf_peer_exchange* x = new (std::nothrow) ppoaf_peer_exchange();
    PPOAF_REQUIRE(x, "peer_exchange_create: out of host memory");
    x->dev.rank = rank; x->dev.n_ranks = n_ranks; x->dev.n4 = bucket_floats / 4;
    x->bytes = kHeaderBytes + 2 * (size_t)bucket_floats * sizeof(float);
    hipError_t e = hipExtMallocWithFlags(&x->base, x->bytes, hipDeviceMallocUncached);
    x->memory_kind = 1;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&x->base, x->bytes, hipDeviceMallocFinegrained);
        x->memory_kind = 2;
    }
    if (e != hipSuccess) {
        set_error("peer_exchange_create: hipExtMallocWithFlags: %s", hipGetErrorString(e));
        delete x;
        return PPOAF_E_LAUNCH;
    }
    void* local = nullptr;
    const size_t local_bytes = 4 * sizeof(long long) + 2 * kXchgMaxGrid * sizeof(double);
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
    x->dev.norm_partials = reinterpret_cast<double*>(x->dev.words + 4);
    x->dev.my_flags = static_cast<long long*>(x->base);
    x->dev.my_slots = reinterpret_cast<float4*>(static_cast<char*>(x->base) + kHeaderBytes);
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
        x->dev.peer_slots[p] = reinterpret_cast<const float4*>(static_cast<char*>(ptr) + kHeaderBytes);
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
    if (blocks > kXchgMaxGrid) blocks = kXchgMaxGrid;    // all workgroups resident: the waits cannot starve a publisher
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
