// Launch-wide hand-off of squared-norm partials inside the fused tail launches (K12: ppo_update_tail.hip, K15:
// mat_update.hip): every workgroup publishes its partial as ONE 16-byte record of two {32 data bits, 32-bit launch tag}
// granules (one `sc1` store; MI355X_MICROARCH.md, data-tagged granules: no flag, no fence, no ordering needed), one wave of
// each workgroup polls all records with `sc1` loads until every tag is this launch's, and adds them lane-strided + xor
// butterfly -- the association of ordered_partial_sum / xchg_ordered_norms in the separate Adam launches, so the clip
// coefficient is bitwise theirs.  All workgroups of such a launch must be resident together (checked by the host); every
// wait is bounded by a wall-clock budget and ends in the control block's error word instead of a hang.
#pragma once
#include "mlp_device.hpp"
#include <cstddef>

namespace ppoaf {

typedef unsigned tail_u32x4 __attribute__((ext_vector_type(4)));
typedef float tail_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTailRecOff = 256;              // byte offset of the records inside the control block
constexpr int kTailMaxRounds = 8;             // a polling lane holds up to 8 records: <= 512 workgroups

struct TailCtl {
    unsigned long long seq;                   // launches completed; a launch tags its records with (seq mod 2^31) + 1
    unsigned error;                           // a wait ran out of its budget: later launches do not wait again
    unsigned pad;
    long long bc_t[2];                        // the Adam step (per network) the corrections below were computed for
    double bc[4];                             // [network][1 - beta1^t, sqrt(1 - beta2^t)]: left by the previous launch's bookkeeping
    unsigned long long stamps[24];            // diagnostic build (-DPPOAF_TAIL_STAMPS): s_memtime per phase of one workgroup
    tail_u32x4 rec[1];                        // [8 * per_xcd]: {q bits 0..31, tag, q bits 32..63, tag}
};
static_assert(offsetof(TailCtl, rec) == kTailRecOff, "record offset");

// diagnostic layer (tools/tail_stamps.py): one macro, nothing in the shipped kernel
#ifdef PPOAF_TAIL_STAMPS
#ifndef PPOAF_TAIL_STAMP_BLOCK
#define PPOAF_TAIL_STAMP_BLOCK 0
#endif
#define TAIL_STAMP(td, k)                                                                          \
    do {                                                                                           \
        if (blockIdx.x == PPOAF_TAIL_STAMP_BLOCK && threadIdx.x == 0) {                            \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            (td).ctl->stamps[k] = t_;                                                              \
        }                                                                                          \
    } while (0)
#else
#define TAIL_STAMP(td, k) do {} while (0)
#endif

struct TailDev {
    TailCtl* ctl;
    long long budget;                         // wall_clock64 ticks (100 MHz)
    int nblk, jobs_a, jobs_c, per_xcd;
};

struct TailCoef { float gs, step_size, bc2_sqrt; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tail_rsrc(const TailDev& td) {
    return __builtin_amdgcn_make_buffer_rsrc(td.ctl, 0, 0xFFFFFFFF, 0x00020000);
}

// one lane: this workgroup's squared-norm partial, visible to every XCD (`sc1`: written through, dropped from this L2)
__device__ __forceinline__ void tail_publish(const TailDev& td, const unsigned tag, const int b, const double q) {
    const unsigned long long qb = (unsigned long long)__double_as_longlong(q);
    const tail_u32x4 r = {(unsigned)qb, tag, (unsigned)(qb >> 32), tag};
    __builtin_amdgcn_raw_buffer_store_b128(r, tail_rsrc(td), (unsigned)(kTailRecOff + 16 * b), 0, 16 /* sc1 */);
}

// wave_sum<double> in its own association (xor 32, 16, 8, 4, 2, 1 -- IEEE addition is commutative, so pairing lane i with
// lane i ^ k is all that matters), with the last four steps on DPP row operations instead of ds_bpermute round trips: after
// the xor-32 and xor-16 steps all four 16-lane rows hold the same vector, row_ror:8 is then xor 8 exactly, the result is
// 8-periodic so row_ror:4 delivers what xor 4 would, and xor 2 / xor 1 are quad permutes (mlp_device.hpp: group16_sum).
template <int CTRL> __device__ __forceinline__ double tail_dpp_d(const double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double tail_wave_sum(double v) {
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    v += tail_dpp_d<kDppRor8>(v);
    v += tail_dpp_d<kDppRor4>(v);
    v += tail_dpp_d<kDppXor2>(v);
    v += tail_dpp_d<kDppXor1>(v);
    return v;
}

// one whole wave: wait until every workgroup's record carries this launch's tag, then the two squared norms in the
// association of xchg_ordered_norms (lane-strided ascending per lane, xor butterfly; an idle or other-network record adds
// +0.0).  Returns false when the wait ran out of its budget (error word set; the sums are then meaningless).
template <int NR>
__device__ __forceinline__ bool tail_gather_n(const TailDev& td, const unsigned tag, double& sq0, double& sq1) {
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = tail_rsrc(td);
    tail_u32x4 r[NR];
    long long budget = td.budget;
    if (__hip_atomic_load(&td.ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) budget = 0;     // broken before: no second wait
    const long long t0 = (long long)wall_clock64();
    bool ok = true;
    for (unsigned polls = 1;; ++polls) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int bb = lane + 64 * k;
            r[k] = tail_u32x4{0u, tag, 0u, tag};
            if (bb < td.nblk) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(kTailRecOff + 16 * bb), 0, 16 /* sc1 */);
        }
        bool all = true;
#pragma unroll
        for (int k = 0; k < NR; ++k) all = all && r[k].y == tag && r[k].w == tag;
        if (__all((int)all)) break;
        if ((polls & 15u) == 0u && (long long)wall_clock64() - t0 > budget) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int bb = lane + 64 * k;
        if (bb < td.nblk) {
            const double q = __longlong_as_double((long long)(((unsigned long long)r[k].z << 32) | (unsigned long long)r[k].x));
            const int job = (bb & 7) * td.per_xcd + (bb >> 3);
            if (job < td.jobs_a) p0 += q; else p1 += q;
        }
    }
    sq0 = tail_wave_sum(p0);
    sq1 = tail_wave_sum(p1);
    if (!ok && lane == 0) __hip_atomic_store(&td.ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
}
__device__ __forceinline__ bool tail_gather(const TailDev& td, const unsigned tag, double& sq0, double& sq1) {
    if (td.nblk <= 192) return tail_gather_n<3>(td, tag, sq0, sq1);          // uniform: a lane holds ceil(nblk / 64) records
    return tail_gather_n<kTailMaxRounds>(td, tag, sq0, sq1);
}

}  // namespace ppoaf
