// K17 device side: the three phases of the peer-mapped gradient exchange (see peer_exchange.hip for the
// protocol).  Shared by the stand-alone exchange kernel and by kernels that fuse the exchange into the
// launch that produces the gradient (ppo_update.hip: slab reduce + exchange in one launch).
#pragma once

#include "common.hpp"

namespace ppoaf {

constexpr int kMaxPeers = PPOAF_PEER_EXCHANGE_MAX_RANKS;
constexpr int kXchgThreads = 256;
constexpr int kXchgMaxGrid = 256;               // stand-alone / slab-reduce exchange launches: one workgroup per CU, all resident
constexpr int kXchgMaxGroups = 512;             // exchange groups an object has flag words for (the fused tail launch of a
                                                // 256-wide critic has 368 job workgroups, two to three per CU)
constexpr size_t kXchgHeaderBytes = (size_t)kXchgMaxGroups * kMaxPeers * 8;   // flag words of one rank [group][peer]; slots follow

struct XchgDev {
    int rank, n_ranks;
    long n4;                                        // float4 elements of one slot
    long long* words;                               // local: [0] exchanges completed, [2] finish count, [3] error
    long long* group_seq;                           // local [kXchgMaxGroups]: sequence number each workgroup has completed
    long long* my_flags;                            // this rank's flag words [group][peer] (polled locally)
    long long* peer_flags[kMaxPeers];               // rank p's flag words (remote store target)
    const float4* peer_slots[kMaxPeers];            // rank p's two slots
    float4* my_slots;
    double* norm_partials;                          // [kXchgMaxGroups][2]
};

// The exchange is organised per WORKGROUP: group g of every rank owns the same elements of the bucket, publishes
// its own flag and waits only for group g of its peers -- no workgroup of a rank waits for another one of the
// same rank, so there is no arrival counter (a memory-side atomic round trip of ~1.5 us) on the path.
// Every launch of an exchange object must use the same number of groups and the same element -> group map.

// Sequence number of the exchange group g performs in this launch (it advances its own word at the end; the
// next launch is ordered behind this one by the stream).
__device__ __forceinline__ long long xchg_sequence(const XchgDev& x, unsigned g) { return x.group_seq[g] + 1; }

// Phase 1 tail: the workgroup's threads have stored its elements of this rank's slot.  Every wave waits until
// its own stores are acknowledged (they have then left the CU and reached the L2 / the uncached memory behind
// it), the barrier collects the waves, and ONE system-scope release fence by the publishing thread writes back
// whatever this XCD's L2 still holds (a system fence in every wave costs 4.5 us more per launch, measured, and
// adds nothing: the waves of a workgroup share the XCD).  Then relaxed posted stores of the sequence number
// into every peer's flag word for (group g, this rank) -- a release store per peer would repeat the write-back.
__device__ __forceinline__ void xchg_publish(const XchgDev& x, long long seq, unsigned g) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        for (int p = 0; p < x.n_ranks; ++p)
            if (p != x.rank)
                __hip_atomic_store(&x.peer_flags[p][g * kMaxPeers + x.rank], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Phase 2: poll the local flag words of group g until every peer's group g has published `seq`, at most
// wait_ticks of the 100 MHz wall clock; on expiry the error word is set and the caller continues (the launch
// must drain).  Lane p of the SECOND wave polls peer p (blockDim.x >= 128), so polling starts while thread 0
// is still publishing.
__device__ __forceinline__ void xchg_wait(const XchgDev& x, long long seq, unsigned g, long long wait_ticks) {
    const int p = (int)threadIdx.x - 64;
    if (p >= 0 && p < x.n_ranks && p != x.rank) {
        // once a wait has expired the exchange is broken for good (the host falls back at the end of the epoch):
        // later launches must not each spend the budget again
        if (__hip_atomic_load(&x.words[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) wait_ticks = 0;
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(&x.my_flags[g * kMaxPeers + p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(1);
            if ((long long)wall_clock64() - t0 > wait_ticks) {
                __hip_atomic_store(&x.words[3], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);        // system scope: drop any cached peer lines
}

// End of the workgroup's exchange: its sequence word moves on (group 0 also keeps the completed count).
__device__ __forceinline__ void xchg_advance(const XchgDev& x, long long seq, unsigned g) {
    if (threadIdx.x == 0) {
        x.group_seq[g] = seq;
        if (g == 0) x.words[0] = seq;
    }
}

// Phase 3 core: element i of the rank-ordered sum; `own` is this rank's contribution (already in registers).
__device__ __forceinline__ float4 xchg_sum(const XchgDev& x, long slot, long i, const float4& own) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p0 = 0; p0 < x.n_ranks; p0 += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = p0 + k;
            if (p >= x.n_ranks) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (p == x.rank) v[k] = own;
            else v[k] = x.peer_slots[p][slot + i];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (p0 + k < x.n_ranks) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    return acc;
}

__device__ __forceinline__ double xchg_sq(const float4& a, float sc) {
    return (double)(a.x * sc) * (a.x * sc) + (double)(a.y * sc) * (a.y * sc) +
           (double)(a.z * sc) * (a.z * sc) + (double)(a.w * sc) * (a.w * sc);
}

// Fixed-association sum of the per-workgroup partials by one wave (all 64 lanes call it): the partials are loaded
// in parallel, one (or a few) per lane, and folded by a butterfly -- the same sum on every rank, without a chain of
// dependent loads.
// (wave_sum: xor-butterfly over the 64 lanes -- the same additions in the same association on every rank and in
// every run, IEEE addition being commutative, so all lanes of all ranks hold the bitwise identical total.)
__device__ __forceinline__ void xchg_ordered_norms(const double* partials, unsigned n_groups, double& n0, double& n1) {
    const unsigned lane = threadIdx.x & 63;
    double p0 = 0.0, p1 = 0.0;
    for (unsigned b = lane; b < n_groups; b += 64) {           // lane-strided, ascending: fixed per-lane order
        p0 += __hip_atomic_load(&partials[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p1 += __hip_atomic_load(&partials[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    n0 = wave_sum(p0);
    n1 = wave_sum(p1);
}

// Phase 3 tail when the norms are wanted by the end of THIS launch: per-workgroup squared-norm partials (q0, q1
// already summed over the workgroup, `group` in [0, n_groups)); the last workgroup to finish adds them in
// workgroup order and writes the norms.  `red_last` is one int of LDS.  (A consumer kernel can instead call
// xchg_ordered_norms on the partials itself and save this launch the counter round trip.)
__device__ __forceinline__ void xchg_finish(const XchgDev& x, unsigned group, unsigned n_groups,
                                            double q0, double q1, bool two_segments, double* norm_out,
                                            int* red_last) {
    if (threadIdx.x == 0) {
        x.norm_partials[2 * group] = q0;
        x.norm_partials[2 * group + 1] = q1;
        __threadfence();
        const unsigned long long prev = atomicAdd(reinterpret_cast<unsigned long long*>(&x.words[2]), 1ull);
        *red_last = prev == (unsigned long long)n_groups - 1ull;
    }
    __syncthreads();
    if (*red_last && threadIdx.x < 64) {
        __threadfence();
        double n0, n1;
        xchg_ordered_norms(x.norm_partials, n_groups, n0, n1);
        if (threadIdx.x == 0) {
            norm_out[0] = n0;
            if (two_segments) norm_out[1] = n1;
            __hip_atomic_store(&x.words[2], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The host object behind ppoaf_peer_exchange_t (opaque in the C ABI).
}  // namespace ppoaf

struct ppoaf_peer_exchange {
    ppoaf::XchgDev dev;
    void* base = nullptr;                           // exported allocation: header + 2 slots
    size_t bytes = 0;
    void* opened[ppoaf::kMaxPeers] = {};
    bool connected = false;
    int memory_kind = 0;                            // 1 uncached, 2 fine-grained
};
