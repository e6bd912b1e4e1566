// K12, fused tail of the split-wgrad chain (round 4): ONE launch after fwd_bwd<SPLIT> forms every weight gradient of
// the mini-batch, the two clip norms, and applies clip + Adam (ppo_policy.py:1032-1055) -- what ppoaf_ppo_update_wgrad +
// ppoaf_ppo_update_adam(compute_norms 3) did in two launches.  Per mini-batch at C2 those two launches were 10.1 us
// (5.05 + 5.01) for ~1 us of arithmetic each: a launch ramp plus a cold round trip for inputs other XCDs had just written
// (profiles/r03_update_pmc.csv: waves parked 64-72 %).  Here
//
//   * every workgroup owns one 16 x 32 piece of one layer's dW (or a network's output segment) END TO END: it requests
//     the optimiser state (p, m, v) of its elements together with its MFMA operands (one cold round trip for both),
//     forms the gradient over all B rows on MFMA (split_wgrad_job's arithmetic and fold order, bit for bit), and applies
//     Adam to exactly those elements -- the gradient never goes through memory on its way to the optimiser;
//   * the one thing every workgroup needs from every other one is the squared norm of the whole gradient (the clip
//     coefficient): each publishes its partial as ONE 16-byte record of two {32 data bits, 32-bit launch tag} granules
//     (one `sc1` store; MI355X_MICROARCH.md, data-tagged granules: no flag, no fence, no ordering needed), and one wave of
//     each workgroup polls all records with `sc1` loads until every tag is this launch's, then adds them lane-strided +
//     xor butterfly -- the association xchg_ordered_norms uses in the Adam launch, so the coefficient, and with it every
//     parameter and moment, is BITWISE what the three-launch chain produces;
//   * a bookkeeping workgroup folds the loss partials into the totals, waits for the same records, and only then
//     advances what the other workgroups read at their start (step counters, the launch tag, the mini-batch cursor).
//
// All workgroups must be resident together (C2: 153 of 256 threads on 256 CUs; 256-wide critics: 369, two to three per
// CU at this kernel's register count -- checked on the host against the occupancy the runtime reports); every wait is
// bounded by a wall-clock budget and ends in the control block's error word instead of a hang (the host then restores
// the epoch's starting state and runs the three-launch chain, fused_update.py).
#include "ppo_update_dev.hpp"
#include "tail_sync.hpp"
#include "peer_exchange_device.hpp"
#include <hip/hip_ext.h>
#include <cstddef>

namespace ppoaf {

constexpr int kTailMaxE = 9;                  // output segment: up to 9 elements per thread (8 x 256 weights + bias + log_std)

// The bias corrections of Adam step t of network `which` (ppo_update_bookkeeping_steps' expressions): the previous launch's
// bookkeeping workgroup left them in the control block while it waited (two double pow: ~2 us of one wave); computed
// here only when the block holds another step's (first launch, restored state).
struct TailBc { long long t; double bc1, bc2s; };
__device__ __forceinline__ TailBc tail_bc_load(const UpdateDev& u, const TailDev& td, const int which) {
    // read at the START of the workgroup: the bookkeeping workgroup rewrites these words only after every workgroup has
    // published its record, i.e. after all of these reads
    TailBc r;
    r.t = td.ctl->bc_t[which];
    r.bc1 = td.ctl->bc[2 * which];
    r.bc2s = td.ctl->bc[2 * which + 1];
    return r;
}

// Wave 0 of the workgroup (all 64 lanes), q = the workgroup's squared-norm partial: publish, wait for everybody's, and
// leave the step's clip / Adam coefficients (ppo_update_adam_kernel's, expression for expression) in s_coef.
__device__ __forceinline__ void tail_sync_wave0(const UpdateDev& u, const TailDev& td, const unsigned tag, const int b, const int which,
                                                const long long t_next, const TailBc& pre, const float lr, const double q,
                                                float* s_coef) {
    if (threadIdx.x == 0) tail_publish(td, tag, b, q);
    double bc1 = pre.bc1, bc2s = pre.bc2s;
    if (pre.t != t_next) {                                    // uniform
        bc1 = 1.0 - pow((double)u.beta1, (double)t_next);
        bc2s = sqrt(1.0 - pow((double)u.beta2, (double)t_next));
    }
    double sq0, sq1;
    tail_gather(td, tag, sq0, sq1);
    if (threadIdx.x == 0) {
        const float total_norm = (float)sqrt(which ? sq1 : sq0);
        float coef = 1.0f;
        if (u.max_norm > 0.f) coef = fminf(u.max_norm / (total_norm + 1e-6f), 1.0f);
        s_coef[0] = u.grad_scale * coef;
        s_coef[1] = (float)((double)lr / bc1);
        s_coef[2] = (float)bc2s;
    }
}

// ---- N > 1: the K17 gradient exchange as a phase of the job (mpi_avg_gradients, utils/mpi_utils.py:89-111, at its call
// sites ppo_policy.py:1035,1048).  Workgroup b of every rank runs the same job, so exchange group b depends only on
// group b of its peers: the job's 16 x 32 sums (+ 16 bias sums) go to THIS rank's slot as 16-byte system-scope stores
// (slot layout = job-major tiles, private to this exchange object: [nblk][528] floats, then the two output segments),
// one lane tells every peer's flag word for (b, this rank), the peers' tiles are read back with 16-byte system-scope
// loads and added IN RANK ORDER (every rank forms the bitwise identical sum), and the job carries on with the summed
// tile: norm partial, the launch-wide wait, clip + Adam.  No fences (a system-scope acquire would drop this XCD's L2
// under the other jobs' operand panels): slot, flag and peer accesses are system-scope accesses that bypass the caches;
// a workgroup's slot stores are acknowledged (vmcnt(0) + barrier) before its flags go out, and a consumer reads a peer's
// tile only after it has seen that peer's flag.  Two slots alternate with the sequence number's parity.
constexpr int kTailTileFloats = 528;          // 16 x 32 tile + 16 bias sums
struct TailXchg { XchgDev x; long long seq, wait_ticks; long seg_base[2]; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tail_slot_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFF, 0x00020000);
}
__device__ __forceinline__ void tail_xchg_publish_wait(const TailXchg& c, const unsigned g) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const XchgDev& x = c.x;
    if (threadIdx.x == 0) {
#pragma unroll                                                // (static indices: the pointer table stays in scalar registers)
        for (int p = 0; p < kMaxPeers; ++p)
            if (p < x.n_ranks && p != x.rank)
                __hip_atomic_store(&x.peer_flags[p][g * kMaxPeers + x.rank], c.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int p = (int)threadIdx.x - 64;
    if (p >= 0 && p < x.n_ranks && p != x.rank) {
        long long budget = c.wait_ticks;
        if (__hip_atomic_load(&x.words[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) budget = 0;      // broken for good
        const long long* flag = &x.my_flags[g * kMaxPeers + p];
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < c.seq) {
            __builtin_amdgcn_s_sleep(1);
            if ((long long)wall_clock64() - t0 > budget) {
                __hip_atomic_store(&x.words[3], c.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}
// float offset `off` (a multiple of 4) inside the slot of this launch's parity: own value out, rank-ordered sum back
__device__ __forceinline__ void tail_xchg_store4(const TailXchg& c, const long off, const tail_f32x4 v) {
    const long byte = ((c.seq & 1) * c.x.n4 * 4 + off) * 4;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tail_u32x4, v), tail_slot_rsrc(c.x.my_slots), (unsigned)byte, 0, 17 /* sc0 sc1 */);
}
__device__ __forceinline__ tail_f32x4 tail_xchg_sum4(const TailXchg& c, const long off, const tail_f32x4 own) {
    const long byte = ((c.seq & 1) * c.x.n4 * 4 + off) * 4;
    tail_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p0 = 0; p0 < kMaxPeers; p0 += 8) {
        if (p0 >= c.x.n_ranks) break;
        tail_f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = p0 + k;
            if (p >= c.x.n_ranks) v[k] = tail_f32x4{0.f, 0.f, 0.f, 0.f};
            else if (p == c.x.rank) v[k] = own;
            else v[k] = __builtin_bit_cast(tail_f32x4, __builtin_amdgcn_raw_buffer_load_b128(tail_slot_rsrc(c.x.peer_slots[p]), (unsigned)byte, 0, 17));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (p0 + k < c.x.n_ranks) acc += v[k];
    }
    return acc;
}
__device__ __forceinline__ void tail_xchg_store1(const TailXchg& c, const long off, const float v) {
    float* mine = reinterpret_cast<float*>(c.x.my_slots) + (c.seq & 1) * c.x.n4 * 4 + off;
    __hip_atomic_store(reinterpret_cast<unsigned*>(mine), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float tail_xchg_sum1(const TailXchg& c, const long off, const float own) {
    const long fo = (c.seq & 1) * c.x.n4 * 4 + off;
    float acc = 0.f;
#pragma unroll
    for (int p0 = 0; p0 < kMaxPeers; p0 += 8) {
        if (p0 >= c.x.n_ranks) break;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = p0 + k;
            if (p >= c.x.n_ranks) v[k] = 0.f;
            else if (p == c.x.rank) v[k] = own;
            else v[k] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(reinterpret_cast<const float*>(c.x.peer_slots[p]) + fo),
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (p0 + k < c.x.n_ranks) acc += v[k];
    }
    return acc;
}

struct TailPmv { float p, m, v; };
__device__ __forceinline__ TailPmv tail_pmv_load(const UpdateDev& u, const long idx, const bool ok) {
    TailPmv r = {0.f, 0.f, 0.f};
    if (ok) { r.p = u.params[idx]; r.m = u.exp_avg[idx]; r.v = u.exp_avg_sq[idx]; }
    return r;
}
// ppo_update_adam_kernel's step on one element (-ffp-contract=off: the same roundings)
__device__ __forceinline__ void tail_adam1(const UpdateDev& u, const long idx, const float g, const TailPmv& s, const TailCoef& c) {
    const float gi = g * c.gs;
    const float m = u.beta1 * s.m + (1.0f - u.beta1) * gi;
    const float v = u.beta2 * s.v + (1.0f - u.beta2) * gi * gi;
    const_cast<float*>(u.params)[idx] = s.p - c.step_size * (m / (sqrtf(v) / c.bc2_sqrt + u.adam_eps));
    u.exp_avg[idx] = m;
    u.exp_avg_sq[idx] = v;
}

// One job of the split-wgrad job list (ppo_update_ws.hip: split_wgrad_job -- same tiles, same operand loads, same fold
// and summation orders), carried through to the optimiser step.  Offsets of p / m / v / G are bucket offsets.
// The 16 x 32 output tile is formed by wave 0 (C layout) and handed to ALL 256 threads through LDS for the optimiser
// step: thread t owns tile elements t and t + 256 (row e / 32, column e % 32: whole 128-byte lines of p / m / v), whose
// state it requested at the start of the job, beside the MFMA operands.
template <int H, bool XCHG>
__device__ __forceinline__ void tail_job(const UpdateDev& u, const TailDev& td, const unsigned tag, const int b, const int which,
                                         const int job, float* sFold /* [3][2][256] + [4][16] */, float* sTile /* [16][32] + [16] */,
                                         double* s_red, float* s_coef, const TailXchg* xc) {
    constexpr int MAXC = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const auto& nd = u.net[which];
    const int in_dim = nd.in_dim, depth = nd.depth, out_dim = nd.out_dim;
    const int B = (int)u.B;
    const long plane = (long)u.sp.Bp * H;
    const long szW0 = ((long)H * in_dim + 3) & ~3L;
    auto offW = [&](int l) -> long { return l == 0 ? 0 : szW0 + H + (long)(l - 1) * ((long)H * H + H); };
    auto offB = [&](int l) -> long {
        return l == 0 ? szW0 : offW(l) + (l < depth ? (long)H * H : (((long)out_dim * H + 3) & ~3L));
    };
    const long nb = nd.offset;                                 // this network's first float in the bucket
    float* G = u.grads + nb;
    // read BEFORE this workgroup publishes (the bookkeeping workgroup rewrites them after everybody has)
    const long long t_next = (long long)u.step_counts[which] + 1;
    const float lr = u.lr[0];
    const TailBc pre = tail_bc_load(u, td, which);
    constexpr int t = H / 16, t2 = (t + 1) / 2;
    const int n_it0 = (in_dim + 15) / 16, p0 = (n_it0 + 1) / 2;
    const int n_hidden = (depth - 1) * t * t2, n_l0 = t * p0;
    const float sc = u.grad_scale;
    double q = 0.0;
    TAIL_STAMP(td, 0);
    if (job < n_hidden + n_l0) {
        int l, ot, itile, n_it;
        if (job < n_hidden) { l = 1 + job / (t * t2); const int jj = job % (t * t2); ot = jj / t2; itile = 2 * (jj % t2); n_it = t; }
        else { l = 0; const int jj = job - n_hidden; ot = jj / p0; itile = 2 * (jj % p0); n_it = n_it0; }
        const bool two = itile + 1 < n_it;                    // uniform per workgroup
        const long ldw = l >= 1 ? H : in_dim;
        // this thread's two tile elements (and, threads 64..79 of the jobs of input piece 0, one bias): optimiser state first
        long eidx[2];
        bool eok[2];
        TailPmv se[2], sb;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tid + 256 * k, row = e >> 5, col = e & 31;
            const int i = itile * 16 + col;
            eok[k] = (col < 16 || two) && i < ldw;
            eidx[k] = nb + offW(l) + (long)(ot * 16 + row) * ldw + i;
            se[k] = tail_pmv_load(u, eidx[k], eok[k]);
        }
        const bool has_b = itile == 0 && tid >= 64 && tid < 80;
        const long bidx = nb + offB(l) + ot * 16 + (tid - 64);
        sb = tail_pmv_load(u, bidx, has_b);
        const long ldx = l >= 1 ? H : 64;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(u.sp.dbuf[which] + (long)l * plane, 0, 0xFFFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            l >= 1 ? u.sp.hbuf[which] + (long)(l - 1) * plane : u.sp.xbuf[which], 0, 0xFFFFFFFF, 0x00020000);
        const unsigned dl = 4u * (unsigned)((lane >> 4) * H + ot * 16 + (lane & 15));
        const unsigned xl = 4u * (unsigned)((lane >> 4) * (int)ldx + itile * 16 + (lane & 15));
        const int nc = (B + 15) >> 4;
        float a[MAXC][4], x0[MAXC][4], x1[MAXC][4];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = wave + 4 * c;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[c][j] = 0.f; x0[c][j] = 0.f; x1[c][j] = 0.f; }
            if (ch < nc) {                                    // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned sd = 4u * (unsigned)((16 * ch + 4 * j) * H), sx = 4u * (unsigned)((16 * ch + 4 * j) * (int)ldx);
                    a[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dl, sd, 0));
                    x0[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl, sx, 0));
                    if (two) x1[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl + 64u, sx, 0));
                }
            }
        }
        TAIL_STAMP(td, 1);
        tail_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (wave + 4 * c < nc) {                          // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x0[c][j], acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x1[c][j], acc1, 0, 0, 0);
                    bsum += a[c][j];
                }
            }
        }
        TAIL_STAMP(td, 2);
        if (wave > 0) {
            *reinterpret_cast<tail_f32x4*>(sFold + (((wave - 1) * 2 + 0) * 64 + lane) * 4) = acc0;
            *reinterpret_cast<tail_f32x4*>(sFold + (((wave - 1) * 2 + 1) * 64 + lane) * 4) = acc1;
        }
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 16) sFold[1536 + wave * 16 + lane] = bsum;
        __syncthreads();
        const int i = itile * 16 + (lane & 15);               // C layout: column = lane & 15, rows 4 (lane >> 4) + r
        float bg = 0.f;
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                acc0 += *reinterpret_cast<const tail_f32x4*>(sFold + ((w * 2 + 0) * 64 + lane) * 4);
                acc1 += *reinterpret_cast<const tail_f32x4*>(sFold + ((w * 2 + 1) * 64 + lane) * 4);
            }
            if (itile == 0 && lane < 16)
                bg = sFold[1536 + lane] + sFold[1536 + 16 + lane] + sFold[1536 + 32 + lane] + sFold[1536 + 48 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                sTile[row * 32 + (lane & 15)] = acc0[r];
                sTile[row * 32 + 16 + (lane & 15)] = acc1[r];
            }
            if (lane < 16) sTile[512 + lane] = bg;
        }
        if (XCHG) {
            // threads 0..127: one float4 of the tile each; 128..131: the bias sums (jobs of input piece 0)
            __syncthreads();
            const bool mine = tid < 128 || (itile == 0 && tid < 132);
            const long off = (long)b * kTailTileFloats + 4 * tid;
            tail_f32x4 own = {0.f, 0.f, 0.f, 0.f};
            if (mine) {
                own = *reinterpret_cast<const tail_f32x4*>(sTile + 4 * tid);
                tail_xchg_store4(*xc, off, own);
            }
            tail_xchg_publish_wait(*xc, (unsigned)b);
            if (mine) *reinterpret_cast<tail_f32x4*>(sTile + 4 * tid) = tail_xchg_sum4(*xc, off, own);
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r;
                    acc0[r] = sTile[row * 32 + (lane & 15)];
                    acc1[r] = sTile[row * 32 + 16 + (lane & 15)];
                }
                if (lane < 16) bg = sTile[512 + lane];
            }
        }
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (i < ldw) q += (double)(acc0[r] * sc) * (acc0[r] * sc);
                if (two && i + 16 < ldw) q += (double)(acc1[r] * sc) * (acc1[r] * sc);
            }
            if (itile == 0 && lane < 16) q += (double)(bg * sc) * (bg * sc);
            TAIL_STAMP(td, 3);
            // block_sum(q) of the chain's wgrad launch: only this wave contributes, and its wave sum + three zeros is exact
            q = tail_wave_sum(q);
            TAIL_STAMP(td, 4);
            tail_sync_wave0(u, td, tag, b, which, t_next, pre, lr, q, s_coef);
        }
        __syncthreads();                                      // s_coef and sTile are visible
        TAIL_STAMP(td, 5);
        const TailCoef cf = {s_coef[0], s_coef[1], s_coef[2]};
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (eok[k]) tail_adam1(u, eidx[k], sTile[tid + 256 * k], se[k], cf);
        if (has_b) tail_adam1(u, bidx, sTile[512 + tid - 64], sb, cf);
        // the gradient bucket itself (nobody in this launch reads it): last, off the step's critical path
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (eok[k]) u.grads[eidx[k]] = sTile[tid + 256 * k];
        if (has_b) u.grads[bidx] = sTile[512 + tid - 64];
        TAIL_STAMP(td, 6);
    } else {
        // output layer (+ log_std): row-block partials -> gradient in block order; a thread keeps its elements
        const long seg_off = offW(depth), seg_len = nd.size - seg_off;
        const float* outpart = u.sp.outpart[which];
        const int n_hb = (B + 15) >> 4;
        TailPmv se[kTailMaxE];
        float ge[kTailMaxE];
#pragma unroll
        for (int k = 0; k < kTailMaxE; ++k) {
            const long idx = tid + (long)kWgradThreads * k;
            se[k] = tail_pmv_load(u, nb + seg_off + idx, idx < seg_len);
            ge[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < kTailMaxE; ++k) {
            const long idx = tid + (long)kWgradThreads * k;
            if (idx < seg_len) {
                float acc = 0.f;
                for (int g0 = 0; g0 < n_hb; g0 += 8) {
                    float pv[8];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) pv[kk] = outpart[(long)(g0 + kk < n_hb ? g0 + kk : 0) * seg_len + idx];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) if (g0 + kk < n_hb) acc += pv[kk];
                }
                ge[k] = acc;
            }
        }
        if (XCHG) {
#pragma unroll
            for (int k = 0; k < kTailMaxE; ++k) {
                const long idx = tid + (long)kWgradThreads * k;
                if (idx < seg_len) tail_xchg_store1(*xc, xc->seg_base[which] + idx, ge[k]);
            }
            tail_xchg_publish_wait(*xc, (unsigned)b);
#pragma unroll
            for (int k = 0; k < kTailMaxE; ++k) {
                const long idx = tid + (long)kWgradThreads * k;
                if (idx < seg_len) ge[k] = tail_xchg_sum1(*xc, xc->seg_base[which] + idx, ge[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < kTailMaxE; ++k) {
            const long idx = tid + (long)kWgradThreads * k;
            if (idx < seg_len) {
                G[seg_off + idx] = ge[k];
                q += (double)(ge[k] * sc) * (ge[k] * sc);
            }
        }
        // block_sum(q) in its own association (wave sums, then the wave sum of the four of them)
        q = tail_wave_sum(q);
        if (lane == 0) s_red[wave] = q;
        __syncthreads();
        if (wave == 0) {
            q = tail_wave_sum(lane < 4 ? s_red[lane] : 0.0);
            tail_sync_wave0(u, td, tag, b, which, t_next, pre, lr, q, s_coef);
        }
        __syncthreads();
        const TailCoef cf = {s_coef[0], s_coef[1], s_coef[2]};
#pragma unroll
        for (int k = 0; k < kTailMaxE; ++k) {
            const long idx = tid + (long)kWgradThreads * k;
            if (idx < seg_len) tail_adam1(u, nb + seg_off + idx, ge[k], se[k], cf);
        }
    }
}

template <int HA, int HC, bool XCHG>
__global__ __launch_bounds__(kWgradThreads) void ppo_update_wgrad_adam_kernel(UpdateDev u, TailDev td, TailXchg xc) {
    __shared__ double s_red[17];
    __shared__ __attribute__((aligned(16))) float s_fold[6 * 256 + 64];
    __shared__ float s_tile[16 * 32 + 16];
    __shared__ float s_coef[4];
    const int b = blockIdx.x;
    const unsigned long long seq = __hip_atomic_load(&td.ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned tag = ((unsigned)seq & 0x7fffffffu) + 1u;              // never 0: a zero-initialised record is never current
    if (b == td.nblk) {
        // bookkeeping: the totals need nobody; what other workgroups read at their start moves only after all have published
        if (threadIdx.x >= 64) return;
        ppo_update_bookkeeping_totals(u);
        // while the others work: the bias corrections of the step being taken (norm_scratch, as ppo_update_bookkeeping_steps
        // leaves them) and of the NEXT step (control block: the next launch's workgroups need not compute them)
        const int w = threadIdx.x & 1;
        const long long t_now = (long long)u.step_counts[w] + 1;
        double c_now[2] = {0.0, 0.0}, c_next[2] = {0.0, 0.0};
        if (threadIdx.x < 2) {
            c_now[0] = 1.0 - pow((double)u.beta1, (double)t_now);
            c_now[1] = sqrt(1.0 - pow((double)u.beta2, (double)t_now));
            c_next[0] = 1.0 - pow((double)u.beta1, (double)(t_now + 1));
            c_next[1] = sqrt(1.0 - pow((double)u.beta2, (double)(t_now + 1)));
        }
        double sq0, sq1;
        tail_gather(td, tag, sq0, sq1);
        if (threadIdx.x < 2) {
            u.step_counts[w] = t_now;
            u.norm_scratch[2 + 2 * w] = c_now[0];
            u.norm_scratch[3 + 2 * w] = c_now[1];
            td.ctl->bc_t[w] = t_now + 1;
            td.ctl->bc[2 * w] = c_next[0];
            td.ctl->bc[2 * w + 1] = c_next[1];
        }
        if (threadIdx.x == 0) {
            __hip_atomic_store(&td.ctl->seq, seq + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (u.cursor_advance) u.cursor[0] += u.cursor_advance;
        }
        return;
    }
    const int job = (b & 7) * td.per_xcd + (b >> 3);          // XCD b % 8 works on one run of the layer-major job list
    const bool live = job < td.jobs_a + td.jobs_c;            // the same on every rank
    if (XCHG && live) xc.seq = xchg_sequence(xc.x, (unsigned)b);
    if (job < td.jobs_a) tail_job<HA, XCHG>(u, td, tag, b, 0, job, s_fold, s_tile, s_red, s_coef, &xc);
    else if (live) tail_job<HC, XCHG>(u, td, tag, b, 1, job - td.jobs_a, s_fold, s_tile, s_red, s_coef, &xc);
    else if (threadIdx.x == 0) tail_publish(td, tag, b, 0.0);
    if (XCHG && live) xchg_advance(xc.x, xc.seq, (unsigned)b);
}

template <int HA, int HC, bool XCHG>
static int tail_launch_as(const UpdateDev& u, const TailDev& td, const TailXchg& xc, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    // every workgroup waits for every other one: all of them must fit on the device at once
    static int per_cu = 0, cus = 0;            // (queried on the first, eager, launch: nothing but the launch inside a stream capture)
    if (per_cu == 0) {
        int n = 0, dev = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(ppo_update_wgrad_adam_kernel<HA, HC, XCHG>),
                                                                   kWgradThreads, 0);
        if (e != hipSuccess) { set_error("ppo_update_wgrad_adam: occupancy query: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        per_cu = n > 0 ? n : -1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    }
    PPOAF_REQUIRE(per_cu > 0 && (long)(td.nblk + 1) <= (long)per_cu * cus,
                  "ppo_update_wgrad_adam: %d workgroups cannot be resident together (%d per CU x %d CUs)", td.nblk + 1, per_cu, cus);
    if (e0 || e1)
        hipExtLaunchKernelGGL((ppo_update_wgrad_adam_kernel<HA, HC, XCHG>), dim3((unsigned)(td.nblk + 1)), dim3(kWgradThreads), 0, s, e0, e1, 0, u, td, xc);
    else
        hipLaunchKernelGGL((ppo_update_wgrad_adam_kernel<HA, HC, XCHG>), dim3((unsigned)(td.nblk + 1)), dim3(kWgradThreads), 0, s, u, td, xc);
    return check_launch("ppo_update_wgrad_adam");
}
template <int HA, int HC>
static int tail_launch(const UpdateDev& u, const TailDev& td, const TailXchg* xc, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (xc) return tail_launch_as<HA, HC, true>(u, td, *xc, s, e0, e1);
    return tail_launch_as<HA, HC, false>(u, td, TailXchg(), s, e0, e1);
}

static int tail_dispatch(const UpdateDev& u, const TailDev& td, const TailXchg* xc, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int ha = u.net[0].H, hc = u.net[1].H;
    if (ha == 32 && hc == 32) return tail_launch<32, 32>(u, td, xc, s, e0, e1);
    if (ha == 64 && hc == 64) return tail_launch<64, 64>(u, td, xc, s, e0, e1);
    if (ha == 128 && hc == 128) return tail_launch<128, 128>(u, td, xc, s, e0, e1);
    if (ha == 256 && hc == 256) return tail_launch<256, 256>(u, td, xc, s, e0, e1);
    if (ha == 128 && hc == 256) return tail_launch<128, 256>(u, td, xc, s, e0, e1);
    if (ha == 64 && hc == 128) return tail_launch<64, 128>(u, td, xc, s, e0, e1);
    set_error("ppo_update_wgrad_adam: hidden widths (actor %d, critic %d) not instantiated", ha, hc);
    return PPOAF_E_INVALID;
}

static int tail_prepare(const ppoaf_ppo_update_args_t* args, void* ctl, double wait_seconds, UpdateDev& u, TailDev& td) {
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    PPOAF_REQUIRE(u.split, "ppo_update_wgrad_adam: args->split_workspace is not set (the tail of the split-wgrad chain)");
    PPOAF_REQUIRE(ctl && (((uintptr_t)ctl) & 63) == 0, "ppo_update_wgrad_adam: control block missing or not 64-byte aligned");
    PPOAF_REQUIRE(wait_seconds > 0.0 && wait_seconds <= 600.0, "ppo_update_wgrad_adam: wait_seconds=%g", wait_seconds);
    td.ctl = reinterpret_cast<TailCtl*>(ctl);
    td.budget = (long long)(wait_seconds * 1.0e8);
    td.jobs_a = split_wgrad_jobs(u.net[0]);
    td.jobs_c = split_wgrad_jobs(u.net[1]);
    td.per_xcd = split_wgrad_per_xcd(u);
    td.nblk = split_wgrad_blocks(u);
    PPOAF_REQUIRE(td.nblk <= 64 * kTailMaxRounds, "ppo_update_wgrad_adam: %d workgroups, a polling wave holds %d records", td.nblk,
                  64 * kTailMaxRounds);
    for (int w = 0; w < 2; ++w)
        PPOAF_REQUIRE(ws_seg_len(u.net[w]) <= (long)kWgradThreads * kTailMaxE, "ppo_update_wgrad_adam: output segment of %ld floats (at most %d)",
                      ws_seg_len(u.net[w]), kWgradThreads * kTailMaxE);
    return PPOAF_OK;
}

// floats of one exchange slot for these shapes: job-major tiles, then the two output segments (each padded to 4)
static long tail_exchange_floats(const UpdateDev& u, long* seg_base) {
    long off = (long)split_wgrad_blocks(u) * kTailTileFloats;
    for (int w = 0; w < 2; ++w) {
        if (seg_base) seg_base[w] = off;
        off += (ws_seg_len(u.net[w]) + 3) & ~3L;
    }
    return off;
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_ppo_update_tail_ctl_bytes(const ppoaf_ppo_update_args_t* args, int64_t* bytes_out) {
    UpdateDev u;
    PPOAF_REQUIRE(args && bytes_out, "ppo_update_tail_ctl_bytes: null argument");
    ppoaf_ppo_update_args_t a = *args;
    a.split_workspace = nullptr;
    const int rc = make_update_dev(&a, u);
    if (rc) return rc;
    *bytes_out = (int64_t)kTailRecOff + 16 * (int64_t)split_wgrad_blocks(u);
    return PPOAF_OK;
}

extern "C" int ppoaf_ppo_update_tail_exchange_floats(const ppoaf_ppo_update_args_t* args, int64_t* floats_out) {
    UpdateDev u;
    PPOAF_REQUIRE(args && floats_out, "ppo_update_tail_exchange_floats: null argument");
    ppoaf_ppo_update_args_t a = *args;
    a.split_workspace = nullptr;
    const int rc = make_update_dev(&a, u);
    if (rc) return rc;
    *floats_out = (int64_t)tail_exchange_floats(u, nullptr);
    return PPOAF_OK;
}

extern "C" int ppoaf_ppo_update_wgrad_adam_timed(const ppoaf_ppo_update_args_t* args, void* ctl, double wait_seconds,
                                                 void* start_event, void* stop_event, ppoaf_stream_t stream) {
    UpdateDev u;
    TailDev td;
    int rc = tail_prepare(args, ctl, wait_seconds, u, td);
    if (rc) return rc;
    return tail_dispatch(u, td, nullptr, (hipStream_t)stream, (hipEvent_t)start_event, (hipEvent_t)stop_event);
}

extern "C" int ppoaf_ppo_update_wgrad_adam(const ppoaf_ppo_update_args_t* args, void* ctl, double wait_seconds, ppoaf_stream_t stream) {
    return ppoaf_ppo_update_wgrad_adam_timed(args, ctl, wait_seconds, nullptr, nullptr, stream);
}

extern "C" int ppoaf_ppo_update_wgrad_adam_exchange(const ppoaf_ppo_update_args_t* args, void* ctl, double wait_seconds,
                                                    ppoaf_peer_exchange_t* xchg, double xchg_wait_seconds, ppoaf_stream_t stream) {
    UpdateDev u;
    TailDev td;
    int rc = tail_prepare(args, ctl, wait_seconds, u, td);
    if (rc) return rc;
    PPOAF_REQUIRE(xchg && xchg->connected, "ppo_update_wgrad_adam_exchange: exchange missing or not connected");
    TailXchg xc;
    xc.x = xchg->dev;
    xc.seq = 0;
    const long need = tail_exchange_floats(u, xc.seg_base);
    PPOAF_REQUIRE(xchg->dev.n4 * 4 >= need, "ppo_update_wgrad_adam_exchange: exchange slots of %ld floats, %ld needed "
                  "(ppoaf_ppo_update_tail_exchange_floats)", (long)xchg->dev.n4 * 4, need);
    PPOAF_REQUIRE(td.nblk <= kXchgMaxGroups, "ppo_update_wgrad_adam_exchange: %d workgroups, one exchange holds %d groups", td.nblk, kXchgMaxGroups);
    PPOAF_REQUIRE(xchg->dev.n_ranks == u.n_ranks || !u.normalize_values, "ppo_update_wgrad_adam_exchange: %d ranks in the exchange, %d in args",
                  xchg->dev.n_ranks, u.n_ranks);
    PPOAF_REQUIRE(xchg_wait_seconds > 0.0 && xchg_wait_seconds <= 600.0, "ppo_update_wgrad_adam_exchange: xchg_wait_seconds=%g", xchg_wait_seconds);
    PPOAF_REQUIRE(xchg->memory_kind != 3, "ppo_update_wgrad_adam_exchange: coarse-grained exchange slots are coherent only through "
                  "fences, which this launch does not use (create the exchange with memory_kind 0, 1 or 2)");
    xc.wait_ticks = (long long)(xchg_wait_seconds * 1.0e8);
    return tail_dispatch(u, td, &xc, (hipStream_t)stream, nullptr, nullptr);
}
