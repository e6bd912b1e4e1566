// K12, two-XCD persistent form: ONE launch runs n_mb consecutive PPO mini-batches (ppo.py:2292-2469) for MLP actor /
// critic networks, single rank.  Each network has its own group of W <= 32 workgroups ("workers"), all on one XCD,
// and one of two decompositions of the mini-batch:
//
//  * ROW-TILED (widths <= 128): the three-launch chain's fwd_bwd body on ceil(B/16) workers, then the slab fold and
//    clip + Adam by parameter-column owners, with the folded gradient held in registers in between.
//  * LAYERED / weight-stationary (256-wide networks).  The row-tiled chain gives every workgroup 16 rows and has it
//    stream the network's whole weight set, twice, per mini-batch: a 256-wide critic workgroup then sits at its own
//    f32-MFMA floor (~20 us).  Layered, the mini-batch is processed LAYER BY LAYER over all of its rows: a phase is a
//    set of independent 64x32 (forward / dgrad) or 64x64 (wgrad, K = all rows) output tiles, each needing one slice
//    of a weight matrix and one slice of an activation matrix, spread over the workers.  Weight gradients are complete
//    sums over the mini-batch where they are produced -- no slabs, no reduce pass -- and the owner of a parameter
//    column applies clip + Adam to it.
//
// What makes that affordable on MI355X: the W workers of a network are all placed on ONE XCD (HW_REG_XCC_ID + a
// ticket, as in the single-XCD persistent form of ppo_update.hip; actor and critic take two different XCDs and never
// wait for each other), so a phase boundary is a flag barrier inside that XCD's L2 (~1.2 us including the first
// dependent load, measured with tools/probes/xcd_persist_probe.hip) and everything a phase hands to the next --
// activations, dz, gradients, the rewritten parameters -- is an L2 hit read with L1-bypassing loads.
//
// Phases of one mini-batch of a LAYERED network (depth d hidden layers; z_l = W_l h_{l-1} + b_l, h_l = act(z_l),
// D_l = dLoss/dz_l; a flag barrier after each line):
//   F(0..d-1)   h_l tiles
//   HEAD        per 16-row block: output layer, distribution head, losses, d out, D_{d-1}, output-layer gradient
//               partials (K6 + K3; same device code as the row-tiled kernel)
//   {DG(m), WG(m+1)} for m = d-1 .. 1   D_{m-1} tiles (dgrad) beside the dW_{m+1} / db_{m+1} tiles (wgrad) of the layer above
//   {WG(1), WG(0), OUT}                  the last weight-gradient tiles, output-layer partials -> gradient, loss
//                                        bookkeeping, squared-norm partials
//   ADAM        clip + Adam on this worker's parameter columns
// (see ws_worker for the fused variants at widths <= 128).  A ROW-TILED network (rt_worker) has three phases: the
// three-launch chain's fwd_bwd body, the slab fold into registers, clip + Adam on those registers.
// Summation orders are fixed (K order inside a tile, K halves, blocks and workers in index order): bitwise reproducible.
#include "ppo_update_rowtile.hpp"
#include "peer_exchange_device.hpp"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace ppoaf {

constexpr int kWsPS = 36;                 // row stride of a 32-column panel slice (conflict-free scalar reads)
constexpr int kWsMaxWorkers = 32;

struct WsCtl {
    unsigned tickets[2], error, done, pad0[28];
    unsigned flags[2][kWsMaxWorkers];     // barrier epoch each worker has reached
    double norm_partials[2][kWsMaxWorkers];
    unsigned long long phase_ticks[16];   // diagnostic build (-DPPOAF_WS_STAMPS): s_memtime sums per phase, actor worker 0
};

#ifdef PPOAF_WS_STAMPS
#define PPOAF_WSTAMP(k)                                                                   \
    do {                                                                                  \
        if (which == PPOAF_WS_STAMP_NET && w == PPOAF_WS_STAMP_WORKER && tid == 0) {      \
            unsigned long long t_;                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
            ctl->phase_ticks[k] += t_ - t_prev;                                           \
            t_prev = t_;                                                                  \
        }                                                                                 \
    } while (0)
#ifndef PPOAF_WS_STAMP_NET
#define PPOAF_WS_STAMP_NET 0
#endif
#ifndef PPOAF_WS_STAMP_WORKER
#define PPOAF_WS_STAMP_WORKER 0
#endif
#define PPOAF_WSTAMP_BEGIN()                                                                          \
    unsigned long long t_prev = 0;                                                                    \
    if (which == PPOAF_WS_STAMP_NET && w == PPOAF_WS_STAMP_WORKER && threadIdx.x == 0)                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory")
#else
#define PPOAF_WSTAMP(k) do {} while (0)
#define PPOAF_WSTAMP_BEGIN() do {} while (0)
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char ppo_update_ws_smem[];

// Flag barrier of the W workers of network `which`, in two halves.  arrive: this worker's stores have reached the L2,
// its flag word says so.  wait: all W flags have reached `epoch`; false when the wait ran out of its budget (error
// word set).  One poll = ONE load instruction: lanes < W read the workers' flag words, lane 63 the error word.
__device__ __forceinline__ void ws_arrive(WsCtl* c, const int which, const int w, const unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) *(volatile unsigned*)&c->flags[which][w] = epoch;   // plain store: the line stays in this XCD's L2
}
__device__ __forceinline__ bool ws_wait(WsCtl* c, const int which, const int W, const unsigned epoch, const long long budget,
                                        int* s_ok) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned* word = lane < W ? &c->flags[which][lane] : &c->error;
        const long long t0 = wall_clock64();
        int ok = 1;
        for (unsigned spin = 1;; ++spin) {
            const unsigned f = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: served by the L2
            if (__all((int)(lane >= W || f >= epoch))) break;
            if (__shfl(f, 63, 64) != 0u || ((spin & 15u) == 0u && wall_clock64() - t0 > budget)) { ok = 0; break; }
        }
        if (lane == 0) {
            if (!ok) *(volatile unsigned*)&c->error = 1u;
            *s_ok = ok;
        }
    }
    __syncthreads();
    return *s_ok != 0;
}
__device__ __forceinline__ bool ws_barrier(WsCtl* c, const int which, const int w, const int W, const unsigned epoch,
                                           const long long budget, int* s_ok) {
    ws_arrive(c, which, w, epoch);
    return ws_wait(c, which, W, epoch, budget, s_ok);
}

// ---- 16x16 output tile of one wave, K in chunks of 16 (four v_mfma_f32_16x16x4_f32 each; the k order inside a
//      chunk is permuted the same way on both operands).  "row" operands: element (i, k) at p[i * stride + k],
//      read as one 16-byte LDS load per chunk; "col" operands: element (k, j) at p[k * kWsPS + j], four scalars.
__device__ __forceinline__ f32x4 ws_mfma_row_row(const float* __restrict__ A, const float* __restrict__ Bm, const int stride,
                                                 const int c0, const int c1, const int lane, const float init) {
    const float* a = A + (lane & 15) * stride + 4 * (lane >> 4);
    const float* b = Bm + (lane & 15) * stride + 4 * (lane >> 4);
    f32x4 acc0 = {init, init, init, init}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int c = c0;
    for (; c + 1 < c1; c += 2) {
        const float4 a0 = *reinterpret_cast<const float4*>(a + 16 * c), b0 = *reinterpret_cast<const float4*>(b + 16 * c);
        const float4 a1 = *reinterpret_cast<const float4*>(a + 16 * c + 16), b1 = *reinterpret_cast<const float4*>(b + 16 * c + 16);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
    }
    if (c < c1) {
        const float4 a0 = *reinterpret_cast<const float4*>(a + 16 * c), b0 = *reinterpret_cast<const float4*>(b + 16 * c);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
    }
    return acc0 + acc1;
}

// C[m][n] = sum_k Ac[k][m] * Bc[k][n]: both operands are 32-column slices with K (the mini-batch rows) down the rows
__device__ __forceinline__ f32x4 ws_mfma_col_col(const float* __restrict__ Ac, const float* __restrict__ Bc,
                                                 const int c0, const int c1, const int lane) {
    const float* a = Ac + (4 * (lane >> 4)) * kWsPS + (lane & 15);
    const float* b = Bc + (4 * (lane >> 4)) * kWsPS + (lane & 15);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int c = c0;
    for (; c + 1 < c1; c += 2) {
        const float* ap = a + (16 * c) * kWsPS;
        const float* bp = b + (16 * c) * kWsPS;
        const float a00 = ap[0], a01 = ap[kWsPS], a02 = ap[2 * kWsPS], a03 = ap[3 * kWsPS];
        const float b00 = bp[0], b01 = bp[kWsPS], b02 = bp[2 * kWsPS], b03 = bp[3 * kWsPS];
        const float a10 = ap[16 * kWsPS], a11 = ap[17 * kWsPS], a12 = ap[18 * kWsPS], a13 = ap[19 * kWsPS];
        const float b10 = bp[16 * kWsPS], b11 = bp[17 * kWsPS], b12 = bp[18 * kWsPS], b13 = bp[19 * kWsPS];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a00, b00, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a10, b10, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a01, b01, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a11, b11, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a02, b02, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a12, b12, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a03, b03, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a13, b13, acc1, 0, 0, 0);
    }
    if (c < c1) {
        const float* ap = a + (16 * c) * kWsPS;
        const float* bp = b + (16 * c) * kWsPS;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], bp[0], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kWsPS], bp[kWsPS], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[2 * kWsPS], bp[2 * kWsPS], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[3 * kWsPS], bp[3 * kWsPS], acc0, 0, 0, 0);
    }
    return acc0 + acc1;
}


// the same with a compile-time chunk count: fully unrolled, so the LDS reads are scheduled ahead of the MFMA chain
template <int NCK>
__device__ __forceinline__ f32x4 ws_mfma_row_row_c(const float* __restrict__ A, const float* __restrict__ Bm, const int stride,
                                                   const int lane, const float init) {
    const float* a = A + (lane & 15) * stride + 4 * (lane >> 4);
    const float* b = Bm + (lane & 15) * stride + 4 * (lane >> 4);
    f32x4 acc0 = {init, init, init, init}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
        const float4 av = *reinterpret_cast<const float4*>(a + 16 * c), bv = *reinterpret_cast<const float4*>(b + 16 * c);
        if (c & 1) {
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc0, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}
template <int NCK>
__device__ __forceinline__ f32x4 ws_mfma_row_col_c(const float* __restrict__ A, const int stride, const float* __restrict__ Bc,
                                                   const int lane) {
    const float* a = A + (lane & 15) * stride + 4 * (lane >> 4);
    const float* b = Bc + (4 * (lane >> 4)) * kWsPS + (lane & 15);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
        const float4 av = *reinterpret_cast<const float4*>(a + 16 * c);
        const float* bp = b + (16 * c) * kWsPS;
        const float b0 = bp[0], b1 = bp[kWsPS], b2 = bp[2 * kWsPS], b3 = bp[3 * kWsPS];
        if (c & 1) {
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b2, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b3, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b2, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b3, acc0, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

// Panel loads, in two halves so that several panels share ONE L2 round trip: issue() requests rows [r0, r0 + nrows)
// of a row-major [*, ld] matrix (16-byte aligned rows), columns [c0, c0 + NCOLS), into registers with L1-bypassing
// 16-byte loads (rows >= row_limit: zero); commit() stores them to dst[r * dstride + c].  NCOLS / 4 is a power of two
// <= 128; MAXROWS bounds nrows (both compile-time: the thread -> (row, column) map is shifts and masks).
template <int NCOLS, int MAXROWS>
struct WsPanel {
    static constexpr int n4 = NCOLS / 4, rpp = kThreadsU / n4, its = (MAXROWS + rpp - 1) / rpp;
    float4 v[its];
};

template <int NCOLS, int MAXROWS>
__device__ __forceinline__ void ws_panel_issue(WsPanel<NCOLS, MAXROWS>& R, const float* __restrict__ src, const long ld,
                                               const int r0, const int nrows, const int row_limit, const int c0, const int tid) {
    using P = WsPanel<NCOLS, MAXROWS>;
    const int r = tid / P::n4, c4 = tid % P::n4;
    const float* sp = src + (long)(r0 + r) * ld + c0 + 4 * c4;
#pragma unroll
    for (int it = 0; it < P::its; ++it) {
        const int rr = r + it * P::rpp;
        R.v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < nrows && r0 + rr < row_limit) R.v[it] = ld4<true>(sp + (long)it * P::rpp * ld);
    }
}
template <int NCOLS, int MAXROWS>
__device__ __forceinline__ void ws_panel_commit(const WsPanel<NCOLS, MAXROWS>& R, float* __restrict__ dst, const int dstride,
                                                const int nrows, const int tid) {
    using P = WsPanel<NCOLS, MAXROWS>;
    const int r = tid / P::n4, c4 = tid % P::n4;
#pragma unroll
    for (int it = 0; it < P::its; ++it) {
        const int rr = r + it * P::rpp;
        if (rr < nrows) *reinterpret_cast<float4*>(dst + rr * dstride + 4 * c4) = R.v[it];
    }
}

// One weight-gradient tile: dW[o0 + 32][i0 + 32] = Dp^T . Xp over the Bk mini-batch rows (both panels [Bk][kWsPS]),
// the K range split over the two wave halves and folded in a fixed order; bias gradient = column sums of Dp when
// `dst_b`.  Returns this thread's contribution to the squared gradient norm (scaled gradients, double).
__device__ __forceinline__ double ws_wgrad_tile(const float* __restrict__ Dp, const float* __restrict__ Xp, float* __restrict__ sRed,
                                                const int Bk, float* __restrict__ dstW, const long ldw, const int o0,
                                                const int i0, const int i_valid, float* __restrict__ dst_b,
                                                const float sc, const int tid, const int wave, const int lane) {
    const int ot2 = wave & 1, it2 = (wave >> 1) & 1, kh = wave >> 2;
    const int nc = Bk >> 4, half = (nc + 1) >> 1;
    const f32x4 acc = ws_mfma_col_col(Dp + 16 * ot2, Xp + 16 * it2, kh ? half : 0, kh ? nc : half, lane);
    if (kh) *reinterpret_cast<f32x4*>(sRed + ((wave & 3) * 64 + lane) * 4) = acc;
    // bias gradient partials: 16 row classes x 32 columns
    float* sB = sRed + 1024;
    if (dst_b) {
        const int col = tid & 31, part = tid >> 5;
        float a = 0.f;
        for (int s = part; s < Bk; s += 16) a += Dp[s * kWsPS + col];
        sB[part * 32 + col] = a;
    }
    __syncthreads();
    double q = 0.0;
    if (!kh) {
        const f32x4 other = *reinterpret_cast<const f32x4*>(sRed + ((wave & 3) * 64 + lane) * 4);
        const int i = i0 + 16 * it2 + (lane & 15);
        if (i < i_valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + 16 * ot2 + 4 * (lane >> 4) + r;
                const float g = acc[r] + other[r];
                dstW[(long)o * ldw + i] = g;
                q += (double)(g * sc) * (g * sc);
            }
        }
    }
    if (dst_b && tid >= 256 && tid < 288) {
        const int col = tid - 256;
        float a = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) a += sB[p * 32 + col];
        dst_b[o0 + col] = a;
        q += (double)(a * sc) * (a * sc);
    }
    return q;
}

constexpr int kWsPSW = 68;                // row stride of a 64-column panel slice

// The same for a 64 x 64 tile (mini-batches of at most 256 rows: both [Bk][kWsPSW] panels fit in LDS): 16 sub-tiles,
// wave w owns output tile (w & 3) x input tiles {2 (w >> 2), 2 (w >> 2) + 1} over the whole K range -- four times the
// MFMA work of the 32 x 32 tile per panel byte and per barrier, and no K fold.
__device__ __forceinline__ double ws_wgrad_tile_wide(const float* __restrict__ Dp, const float* __restrict__ Xp, float* __restrict__ sRed,
                                                     const int Bk, float* __restrict__ dstW, const long ldw, const int o0,
                                                     const int i0, const int i_valid, float* __restrict__ dst_b,
                                                     const float sc, const int tid, const int wave, const int lane) {
    const int ot4 = wave & 3, ip = (wave >> 2) * 2;
    const float* a = Dp + (4 * (lane >> 4)) * kWsPSW + 16 * ot4 + (lane & 15);
    const float* b = Xp + (4 * (lane >> 4)) * kWsPSW + 16 * ip + (lane & 15);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int nc = Bk >> 4;
#pragma unroll 2
    for (int c = 0; c < nc; ++c) {
        const float* ap = a + (16 * c) * kWsPSW;
        const float* bp = b + (16 * c) * kWsPSW;
        const float a0 = ap[0], a1 = ap[kWsPSW], a2 = ap[2 * kWsPSW], a3 = ap[3 * kWsPSW];
        const float b00 = bp[0], b01 = bp[kWsPSW], b02 = bp[2 * kWsPSW], b03 = bp[3 * kWsPSW];
        const float b10 = bp[16], b11 = bp[kWsPSW + 16], b12 = bp[2 * kWsPSW + 16], b13 = bp[3 * kWsPSW + 16];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b00, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b10, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b01, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b11, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b02, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b12, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b03, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b13, acc1, 0, 0, 0);
    }
    // bias gradient partials: 8 row classes x 64 columns
    float* sB = sRed;
    if (dst_b) {
        const int col = tid & 63, part = tid >> 6;
        float s8 = 0.f;
        for (int s = part; s < Bk; s += 8) s8 += Dp[s * kWsPSW + col];
        sB[part * 64 + col] = s8;
        __syncthreads();
    }
    double q = 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4 acc = t ? acc1 : acc0;
        const int i = i0 + 16 * (ip + t) + (lane & 15);
        if (i < i_valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + 16 * ot4 + 4 * (lane >> 4) + r;
                dstW[(long)o * ldw + i] = acc[r];
                q += (double)(acc[r] * sc) * (acc[r] * sc);
            }
        }
    }
    if (dst_b && tid < 64) {
        float s8 = 0.f;
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) s8 += sB[p8 * 64 + tid];
        dst_b[o0 + tid] = s8;
        q += (double)(s8 * sc) * (s8 * sc);
    }
    return q;
}

// x: the K17 exchange of an N > 1 launch (xchg_ranks = its rank count, 0 on a single rank: no exchange phase at all)
struct WsArgs { UpdateDev u; WsDev ws; XchgDev x; long long xchg_ticks; int xchg_ranks, xchg_fences; };
typedef const WsArgs __attribute__((address_space(4))) KWsArgs;

// ---- K17 inside the persistent launch (N > 1 ranks; the reference's per-mini-batch mpi_avg_gradients,
//      utils/mpi_utils.py:89-111 at ppo.py:2443-2448).  Worker w of network `which` on every rank owns the same float4
//      columns of the bucket (idx = lo4 + (c W + w) 512 + tid), so exchange group g = which * 32 + w of every rank
//      depends only on group g of its peers, exactly as in peer_exchange.hip: own columns -> this rank's slot, flag
//      words into every peer, poll the local flag words, peers' columns added in rank order (bitwise the same sum on
//      every rank).  Differences from the stand-alone launch, because this launch lives on for a whole epoch and the
//      XCD's L2 holds its working set:
//        * slot stores, flag stores, flag polls and peer reads are all system-scope (sc0 sc1) atomic accesses of 8
//          bytes: each is coherent by itself (written through / fetched from memory), so no cache invalidate
//          (buffer_inv sc1 would also drop this XCD's L2 lines: measured +14 us per phase, l1_inv_probe.hip) and no
//          L2 write-back is needed around them; xchg_fences = 1 adds the formal system-scope release / acquire
//          fences anyway (exchange slots in ordinary coarse-grained memory need them);
//        * the group's sequence number lives in a register for the launch (read at start, written back at the end).
template <typename KA>
__device__ __forceinline__ void ws_xchg_store(KA* ka, const long at, const float4& v) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(ka->x.my_slots + at);
    const unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
    const unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
    __hip_atomic_store(d, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(d + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float4 ws_xchg_load(const float4* src) {
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
    const unsigned long long lo = __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long hi = __hip_atomic_load(s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}
// all of the workgroup's slot stores are acknowledged -> this rank's sequence number into every peer's flag word for
// (group g, this rank); then the second wave polls the local flag words (lane p: peer p), bounded by the wall clock
template <typename KA>
__device__ __forceinline__ void ws_xchg_publish_wait(KA* ka, const long long seq, const unsigned g) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int R = ka->x.n_ranks, me = ka->x.rank;
    if (threadIdx.x == 0) {
        if (ka->xchg_fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        for (int p = 0; p < R; ++p)
            if (p != me)
                __hip_atomic_store(&ka->x.peer_flags[p][g * kMaxPeers + me], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int p = (int)threadIdx.x - 64;
    if (p >= 0 && p < R && p != me) {
        long long* words = ka->x.words;
        long long budget = ka->xchg_ticks;
        if (__hip_atomic_load(&words[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) budget = 0;   // broken for good
        const long long* flag = &ka->x.my_flags[g * kMaxPeers + p];
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(1);
            if ((long long)wall_clock64() - t0 > budget) {
                __hip_atomic_store(&words[3], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    if (ka->xchg_fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// column `at` (slot offset included) of the rank-ordered sum; `own` = this rank's contribution
template <typename KA>
__device__ __forceinline__ float4 ws_xchg_sum(KA* ka, const long at, const float4& own) {
    const int R = ka->x.n_ranks, me = ka->x.rank;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p0 = 0; p0 < R; p0 += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = p0 + k;
            if (p >= R) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (p == me) v[k] = own;
            else v[k] = ws_xchg_load(ka->x.peer_slots[p] + at);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (p0 + k < R) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    return acc;
}

// dgrad tile product with the weight matrix held ROW-major in LDS (rows = k, row stride PSTR): the tail phase reads the
// same [H][H + 4] copy of W forwards (row_row) and backwards (this)
template <int NCK, int PSTR>
__device__ __forceinline__ f32x4 ws_mfma_row_colT_c(const float* __restrict__ A, const int stride, const float* __restrict__ Bc,
                                                    const int lane) {
    const float* a = A + (lane & 15) * stride + 4 * (lane >> 4);
    const float* b = Bc + (4 * (lane >> 4)) * PSTR + (lane & 15);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
        const float4 av = *reinterpret_cast<const float4*>(a + 16 * c);
        const float* bp = b + (16 * c) * PSTR;
        const float b0 = bp[0], b1 = bp[PSTR], b2 = bp[2 * PSTR], b3 = bp[3 * PSTR];
        if (c & 1) {
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b2, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b3, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b2, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b3, acc0, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

// One worker of network WHICH (hidden width H) for the whole launch, layered mode.  Everything the body derives from
// the kernel arguments or the lane id is re-derived per mini-batch from laundered copies (see ppo_update.hip,
// persistent form): hoisted out of the mini-batch loop it would all be live at once and spill.
//
// Phase schedule of one mini-batch (d = depth; a barrier after each line):
//   generic (H = 256):      F(0) | F(1) | .. | F(d-1) | HEAD | {DG(l), WG(l)} for l = d-1 .. 1 (OUT with the first) |
//                           {WG(0), bookkeeping} | ADAM
//   H <= 128 and d >= 2:    the last hidden layer's weights fit in LDS next to a 16-row block, so F(d-1), HEAD and
//                           DG(d-1) run row-locally in ONE phase ("tail"): F(0) | .. | F(d-2) | TAIL |
//                           {DG(m), WG(m+1)} for m = d-2 .. 1 | {WG(1), WG(0), OUT, bookkeeping} | ADAM
//   in_dim <= 16, d >= 3 (or generic d >= 2): F(0) is computed on the fly inside the F(1) jobs (K = 16: four MFMAs
//                           per tile); the jobs of column tile 0 also publish h_0 for the backward pass
// DG(l): D_{l-1} = (D_l . W_l) * act'(h_{l-1});  WG(l): dW_l = D_l^T h_{l-1}, db_l;  OUT: output-layer partials -> gradient.
template <int WHICH, int H>
__device__ __forceinline__ void ws_worker(KWsArgs* ka0, WsCtl* ctl, const int w, const int n_mb, const long long budget,
                                          int* s_ok, double* s_red, double* s_norm, float* s_step) {
    constexpr int which = WHICH;
    constexpr int HS = H + 4, NCH = H / 16, nct = H / 32;
    constexpr bool kTail = H <= 128;
    // ---- LDS carve
    float* smem = reinterpret_cast<float*>(ppo_update_ws_smem);
    int* sRow = reinterpret_cast<int*>(smem);                 // [512] dataset row of every mini-batch row (-1: dead)
    int* sDi = sRow + 512;                                    // [512] where its inputs are read from (-1: dead)
    float* sT0 = smem + 1024;                                 // [512] per row: advantage (actor) / rewards-to-go (critic)
    float* sT1 = sT0 + 512;                                   // [512] per row: old log-prob (actor)
    float* sMisc = smem + 2048;                               // [16]
    float* sRowF = sMisc + 16;                                // [3][16]
    float* sActF = sRowF + 48;                                // [16][8]
    float* sOut = sActF + 128;                                // [16][16]
    float* sDOut = sOut + 256;                                // [16][16]
    float* sRed = sDOut + 256;                                // [1024 + 512] wgrad K-half fold, bias partials
    float* sP = sRed + 1536;                                  // panels

    const long cursor0 = ka0->u.cursor[0];                    // rewritten only after both networks have finished
    const int W = ka0->ws.W;
    unsigned epoch = 0;
    const unsigned xg = (unsigned)(which * kWsMaxWorkers + w);                // exchange group of this worker (N > 1)
    long long xseq = ka0->xchg_ranks > 0 ? ka0->x.group_seq[xg] : 0;
    PPOAF_WSTAMP_BEGIN();

    for (int it = 0; it < n_mb; ++it) {
        KWsArgs* ka = ka0;
        asm volatile("" : "+s"(ka));
        const auto& u = ka->u;
        const auto& ws = ka->ws;
        int tid_l = threadIdx.x;
        asm volatile("" : "+v"(tid_l));
        __builtin_assume(tid_l >= 0 && tid_l < kThreadsU);
        const int tid = tid_l, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

        const auto& nd = u.net[which];
        const int in_dim = nd.in_dim, depth = nd.depth, out_dim = nd.out_dim, act = nd.act;
        const int in_pad = (in_dim + 15) & ~15;
        const long B = u.B;
        const int Bp = ws.Bp, Bk = ((int)B + 15) & ~15;
        const int n_rt = Bp >> 6, n_hb = ((int)B + 15) >> 4;
        const bool tail = kTail && depth >= 2;
        const bool fuse0 = in_pad == 16 && depth >= (tail ? 3 : 2);
        const long szW0 = ((long)H * in_dim + 3) & ~3L;
        auto offW = [&](int l) -> long { return l == 0 ? 0 : szW0 + H + (long)(l - 1) * ((long)H * H + H); };
        auto offB = [&](int l) -> long {
            return l == 0 ? szW0 : offW(l) + (l < depth ? (long)H * H : (((long)out_dim * H + 3) & ~3L));
        };
        const long seg_off = offW(depth), seg_len = nd.size - seg_off;
        const long relB = ((long)out_dim * H + 3) & ~3L, relLS = relB + ((out_dim + 3) & ~3);
        const float* P = u.params + nd.offset;
        float* G = u.grads + nd.offset;
        float* hbuf = ws.hbuf[which];
        float* dbuf = ws.dbuf[which];
        float* outpart = ws.outpart[which];
        float* xbuf = ws.xbuf[which];
        const long plane = (long)Bp * H;
        const float* xsrc = which == 0 ? u.obs : u.critic_obs;
        const long mb = cursor0 + it;
        double sumsq = 0.0;

        // ---- row table of a mini-batch: dataset row, where its inputs are read from, and the per-row scalars of the loss
        //      (index chains and cold reads: built for mini-batch it + 1 while the Adam barrier of mini-batch it
        //      propagates -- the table is dead after the HEAD phase -- so that none of it sits on the critical path)
        auto build_row_table = [&](const long mbi) {
            const long bs = mbi * u.batch_stride;
            for (int s = tid; s < Bp; s += kThreadsU) {
                int row = -1;
                long di = -1;
                if (s < B) {
                    const long p = u.perm[bs + s];
                    if (p >= 0 && p < u.n_rows) row = u.row_map ? u.row_map[p] : (int)p;
                    di = u.pregathered ? bs + s : row;
                }
                const bool live = row >= 0;
                const long dl = live ? di : 0;
                const float t0 = which == 0 ? u.adv[dl] : u.rtg[dl];
                const float t1 = which == 0 ? u.old_lp[dl] : 0.f;
                sRow[s] = row;
                sDi[s] = live ? (int)di : -1;
                sT0[s] = live ? t0 : 0.f;
                sT1[s] = live ? t1 : 0.f;
            }
            __syncthreads();
        };
        if (it == 0) build_row_table(mb);

        // ================================================================ job bodies
        // forward layer 0 alone: h_0[64 rows][32 columns] tile, K = in_pad (<= 64): 16 threads per row, no divisions
        auto f0_job = [&](const int j) {
            const int AS = in_pad + 4, nck = in_pad >> 4;
            float* sA = sP;
            float* sW = sP + 64 * AS;
            const int rt = j / nct, ct = j % nct;
            const int o = ct * 32 + 16 * (wave >> 2) + (lane & 15);
            const float bias = ld1<true>(P + offB(0) + o);
            const int r16 = tid >> 4, kq = tid & 15;
            float xa[8], wv[4];
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int di = sDi[rt * 64 + r16 + 32 * ps];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = kq + 16 * i;
                    const bool ok = di >= 0 && k < in_dim;
                    const float v = xsrc[(long)(di >= 0 ? di : 0) * in_dim + (k < in_dim ? k : 0)];
                    xa[ps * 4 + i] = ok ? v : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kq + 16 * i;
                const float v = ld1<true>(P + (long)(ct * 32 + r16) * in_dim + (k < in_dim ? k : 0));
                wv[i] = k < in_dim ? v : 0.f;
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nck) sA[(r16 + 32 * ps) * AS + kq + 16 * i] = xa[ps * 4 + i];
                    if (ct == 0) xbuf[(long)(rt * 64 + r16 + 32 * ps) * 64 + kq + 16 * i] = i < nck ? xa[ps * 4 + i] : 0.f;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nck) sW[r16 * AS + kq + 16 * i] = wv[i];
            __syncthreads();
            const f32x4 acc = ws_mfma_row_row(sA + 16 * (wave & 3) * AS, sW + 16 * (wave >> 2) * AS, AS, 0, nck, lane, bias);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = rt * 64 + 16 * (wave & 3) + 4 * (lane >> 4) + r;
                if (s < B) hbuf[(long)s * H + o] = act_fwd(acc[r], act);
            }
            __syncthreads();
        };

        // forward hidden layer l: h_l[64 rows][32 columns] tile; with x0 (in_dim <= 16) h_0 of the 64 rows is computed
        // here from the inputs instead of read back (and published by the jobs of column tile 0)
        auto f_job = [&](const int l, const int j, const bool x0) {
            float* sA = sP;                                   // [64][HS]
            float* sW = sP + 64 * HS;                         // [32][HS]
            float* sX0 = sW + 32 * HS;                        // [64][20]   x0 only
            float* sW0 = sX0 + 64 * 20;                       // [H][20]
            float* sB0 = sW0 + H * 20;                        // [H]
            const float* Wl = P + offW(l);
            float* hout = hbuf + (long)l * plane;
            const int rt = j / nct, ct = j % nct;
            const int o = ct * 32 + 16 * (wave >> 2) + (lane & 15);
            const float bias = ld1<true>(P + offB(l) + o);
            WsPanel<H, 32> rw;
            ws_panel_issue(rw, Wl, H, ct * 32, 32, H, 0, tid);
            if (!x0) {
                WsPanel<H, 64> ra;
                ws_panel_issue(ra, hbuf + (long)(l - 1) * plane, H, rt * 64, 64, (int)B, 0, tid);
                ws_panel_commit(ra, sA, HS, 64, tid);
            } else {
                // x rows [64][16] (2 values per thread), W_0 [H][16], b_0 [H]
                const int r8 = tid >> 3, k2 = (tid & 7) * 2;
                const int di = sDi[rt * 64 + r8];
                const float* xp = xsrc + (long)(di >= 0 ? di : 0) * in_dim;
                const float x_a = xp[k2 < in_dim ? k2 : 0], x_b = xp[k2 + 1 < in_dim ? k2 + 1 : 0];
                float w0v[H * 16 / kThreadsU];
#pragma unroll
                for (int i = 0; i < H * 16 / kThreadsU; ++i) {
                    const int e = tid + i * kThreadsU, r = e >> 4, k = e & 15;
                    w0v[i] = ld1<true>(P + (long)r * in_dim + (k < in_dim ? k : 0));
                }
                const float b0v = tid < H ? ld1<true>(P + offB(0) + tid) : 0.f;
                const float xa0 = (di >= 0 && k2 < in_dim) ? x_a : 0.f, xb0 = (di >= 0 && k2 + 1 < in_dim) ? x_b : 0.f;
                sX0[r8 * 20 + k2] = xa0;
                sX0[r8 * 20 + k2 + 1] = xb0;
                if (ct == 0) {
                    float* xr = xbuf + (long)(rt * 64 + r8) * 64;
                    xr[k2] = xa0; xr[k2 + 1] = xb0;
#pragma unroll
                    for (int kk = 16; kk < 64; kk += 16) { xr[kk + k2] = 0.f; xr[kk + k2 + 1] = 0.f; }
                }
#pragma unroll
                for (int i = 0; i < H * 16 / kThreadsU; ++i) {
                    const int e = tid + i * kThreadsU, r = e >> 4, k = e & 15;
                    sW0[r * 20 + k] = k < in_dim ? w0v[i] : 0.f;
                }
                if (tid < H) sB0[tid] = b0v;
                __syncthreads();
                for (int t = wave; t < 4 * NCH; t += kNW) {           // 4 row tiles x NCH column tiles of h_0
                    const int rt4 = t & 3, c16 = t >> 2;
                    const int oc = 16 * c16 + (lane & 15);
                    const f32x4 a0 = ws_mfma_row_row(sX0 + 16 * rt4 * 20, sW0 + 16 * c16 * 20, 20, 0, 1, lane, sB0[oc]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int sl = 16 * rt4 + 4 * (lane >> 4) + r, s = rt * 64 + sl;
                        const float hv = act_fwd(a0[r], act);
                        sA[sl * HS + oc] = hv;
                        if (ct == 0 && s < B) hbuf[(long)s * H + oc] = hv;
                    }
                }
            }
            ws_panel_commit(rw, sW, HS, 32, tid);
            __syncthreads();
            const f32x4 acc = ws_mfma_row_row_c<NCH>(sA + 16 * (wave & 3) * HS, sW + 16 * (wave >> 2) * HS, HS, lane, bias);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = rt * 64 + 16 * (wave & 3) + 4 * (lane >> 4) + r;
                if (s < B) hout[(long)s * H + o] = act_fwd(acc[r], act);
            }
            __syncthreads();
        };

        // HEAD on one 16-row block.  with_tail: the block's h_{d-1} is first computed from h_{d-2} with W_{d-1} held in
        // LDS, and D_{d-2} is produced from D_{d-1} with the same copy afterwards.
        auto head_job = [&](const int g, const bool with_tail) {
            float* sH = sP;                                   // [16][HS]  h_{d-1}
            float* sWout = sH + 16 * HS;                      // [8][H]
            float* sBout = sWout + 8 * H;                     // [16]
            float* sHp = sBout + 16;                          // [16][HS]  h_{d-2}            (tail)
            float* sDl = sHp + 16 * HS;                       // [16][HS]  D_{d-1}            (tail)
            float* sBl = sDl + 16 * HS;                       // [H]       b_{d-1}            (tail)
            float* sWl = sBl + H;                             // [H][HS]   W_{d-1}            (tail)
            const float* hlast = hbuf + (long)(depth - 1) * plane;
            float* dlast = dbuf + (long)(depth - 1) * plane;
            WsPanel<H, 16> rh;
            WsPanel<H, kTail ? H : 16> rwl;
            float bl = 0.f;
            if (with_tail) {
                ws_panel_issue(rh, hbuf + (long)(depth - 2) * plane, H, g * 16, 16, (int)B, 0, tid);
                ws_panel_issue(rwl, P + offW(depth - 1), H, 0, H, H, 0, tid);
                if (tid < H) bl = ld1<true>(P + offB(depth - 1) + tid);
            } else {
                ws_panel_issue(rh, hlast, H, g * 16, 16, (int)B, 0, tid);
            }
            float wo[(8 * H + kThreadsU - 1) / kThreadsU];
#pragma unroll
            for (int i = 0; i < (8 * H + kThreadsU - 1) / kThreadsU; ++i) {
                const int e = tid + i * kThreadsU;
                wo[i] = ld1<true>(P + offW(depth) + (e < out_dim * H ? e : 0));
            }
            float bo = 0.f;
            if (tid < out_dim) bo = ld1<true>(P + offB(depth) + tid);
            if (tid < kRows) {
                const int s = g * kRows + tid;
                const int di = sDi[s];
                if (di >= 0 && which == 0) {
                    if (u.head_kind == PPOAF_HEAD_CATEGORICAL)
                        reinterpret_cast<int*>(sActF)[tid * 8] = (int)reinterpret_cast<const int64_t*>(u.raw_actions)[di];
                    else
                        for (int d = 0; d < out_dim; ++d)
                            sActF[tid * 8 + d] = reinterpret_cast<const float*>(u.raw_actions)[(long)di * out_dim + d];
                }
                sRowF[tid] = which == 0 ? sT0[s] : 0.f; sRowF[16 + tid] = sT1[s]; sRowF[32 + tid] = which == 1 ? sT0[s] : 0.f;
            }
            if (tid == 64) {                                  // mini-batch statistics (as the row-tiled kernel's S0)
                if (which == 0) {
                    float mean_f = 0.f, std_f = 1.f;
                    if (u.normalize_adv) {                    // ppo.py:2326-2333, from the per-epoch table
                        const double* rec = u.adv_records + mb * 3;
                        mean_f = (float)rec[1];
                        std_f = (float)sqrt(rec[2] / (rec[0] - 1.0));
                    }
                    sMisc[0] = mean_f; sMisc[1] = std_f;
                } else {
                    const int slot = (int)(mb & 1);
                    float m = ld1<true>(u.vn_mean + slot), v = ld1<true>(u.vn_var + slot);
                    double cnt = ld1<true>(u.vn_count + slot);
                    if (u.normalize_values) {                 // Chan merge of the per-rank records + utils/stats.py:73-94
                        double n = 0.0, bm = 0.0, M2 = 0.0;
                        for (int r = 0; r < u.n_ranks; ++r) {
                            const double* rec = u.vn_records + (mb * u.n_ranks + r) * 3;
                            const double nb = rec[0];
                            if (nb <= 0.0) continue;
                            const double d = rec[1] - bm, nn = n + nb;
                            bm += d * (nb / nn);
                            M2 += rec[2] + d * d * n * nb / nn;
                            n = nn;
                        }
                        if (n > 0.0) {
                            const float batch_mean = (float)bm, batch_var = (float)(M2 / n);
                            const float delta = batch_mean - m;
                            const double new_count = cnt + n;
                            const float new_mean = (float)((double)m + (double)delta * (n / new_count));
                            const double m_2 = (double)v * cnt + (double)batch_var * n +
                                               (double)(delta * delta) * cnt * n / (cnt + n);
                            m = new_mean; v = (float)(m_2 / (cnt + n)); cnt = new_count;
                        }
                    }
                    sMisc[2] = m; sMisc[3] = v;
                    if (g == 0) { u.vn_mean[slot ^ 1] = m; u.vn_var[slot ^ 1] = v; u.vn_count[slot ^ 1] = cnt; }
                }
            }
#pragma unroll
            for (int i = 0; i < (8 * H + kThreadsU - 1) / kThreadsU; ++i) {
                const int e = tid + i * kThreadsU;
                if (e < out_dim * H) sWout[e] = wo[i];
            }
            if (tid < out_dim) sBout[tid] = bo;
            if (with_tail) {
                ws_panel_commit(rh, sHp, HS, 16, tid);
                ws_panel_commit(rwl, sWl, HS, H, tid);
                if (tid < H) sBl[tid] = bl;
                __syncthreads();
                for (int t = wave; t < NCH; t += kNW) {               // h_{d-1}[16][H]: one 16-column tile per wave
                    const int oc = 16 * t + (lane & 15);
                    const f32x4 acc = ws_mfma_row_row_c<NCH>(sHp, sWl + 16 * t * HS, HS, lane, sBl[oc]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sH[(4 * (lane >> 4) + r) * HS + oc] = act_fwd(acc[r], act);
                }
            } else {
                ws_panel_commit(rh, sH, HS, 16, tid);
            }
            __syncthreads();
            if (tid < 256) {                                  // output layer: 16 lanes per row
                const int s = tid >> 4, part = tid & 15;
                for (int k = 0; k < out_dim; ++k) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < H; i += 16) acc = fmaf(sH[s * HS + part + i], sWout[k * H + part + i], acc);
                    acc = group16_sum(acc);
                    if (part == 0) sOut[s * kMaxOut + k] = acc + sBout[k];
                }
            }
            __syncthreads();
            if (wave == 0)
                ppo_head_loss<true>(u, which, g, out_dim, P + nd.log_std_off, sRow + g * kRows, sRowF, sMisc, sActF, sOut,
                                    sDOut, lane, B);
            __syncthreads();
            float* op = outpart + (long)g * seg_len;
            if (tid < H) {                                    // dW_out partial of this row block
                const int i = tid;
                float h[kRows];
#pragma unroll
                for (int s = 0; s < kRows; ++s) h[s] = sH[s * HS + i];
                for (int k = 0; k < out_dim; ++k) {
                    float acc = 0.f;
#pragma unroll
                    for (int s = 0; s < kRows; ++s) acc = fmaf(sDOut[s * kMaxOut + k], h[s], acc);
                    op[(long)k * H + i] = acc;
                }
            }
            if (tid >= 256 && tid < 256 + ((out_dim + 3) & ~3)) {
                const int k = tid - 256;
                float acc = 0.f;
                if (k < out_dim)
                    for (int s = 0; s < kRows; ++s) acc += sDOut[s * kMaxOut + k];
                op[relB + k] = acc;
            }
            if (which == 0 && u.head_kind == PPOAF_HEAD_GAUSSIAN && tid >= 320 && tid < 320 + ((out_dim + 3) & ~3)) {
                const int d = tid - 320;
                float acc = 0.f;
                if (d < out_dim)
                    for (int s = 0; s < kRows; ++s) acc += sOut[s * kMaxOut + 8 + d];
                op[relLS + d] = acc;
            }
#pragma unroll
            for (int idx = tid; idx < kRows * H; idx += kThreadsU) {              // D_{depth-1} rows of this block
                const int s = idx / H, i = idx % H;
                float acc = 0.f;
                for (int k = 0; k < out_dim; ++k) acc = fmaf(sDOut[s * kMaxOut + k], sWout[k * H + i], acc);
                const int gs = g * kRows + s;
                const float dz = acc * act_bwd(sH[s * HS + i], act);
                if (gs < B) dlast[(long)gs * H + i] = dz;
                if (with_tail) sDl[s * HS + i] = dz;
            }
            __syncthreads();
            if (with_tail) {
                float* dprev = dbuf + (long)(depth - 2) * plane;
                for (int t = wave; t < NCH; t += kNW) {               // D_{d-2}[16][H] = (D_{d-1} . W_{d-1}) * act'(h_{d-2})
                    const int ic = 16 * t + (lane & 15);
                    const f32x4 acc = ws_mfma_row_colT_c<NCH, HS>(sDl, HS, sWl + 16 * t, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int sl = 4 * (lane >> 4) + r, gs = g * kRows + sl;
                        if (gs < B) dprev[(long)gs * H + ic] = acc[r] * act_bwd(sHp[sl * HS + ic], act);
                    }
                }
                __syncthreads();
            }
        };

        // DG(l), l >= 1: D_{l-1}[64 rows][32 cols] = (D_l . W_l[:, cols]) * act'(h_{l-1})
        auto dgrad_job = [&](const int l, const int j) {
            const float* Dl = dbuf + (long)l * plane;
            const float* Wl = P + offW(l);
            const int rt = j / nct, ct = j % nct;
            float* sA = sP;                                   // D_l rows [64][HS]
            float* sWc = sP + 64 * HS;                        // W_l[:, ct*32 .. +32]  [H][kWsPS]
            const float* hin = hbuf + (long)(l - 1) * plane;
            float* dout = dbuf + (long)(l - 1) * plane;
            const int i = ct * 32 + 16 * (wave >> 2) + (lane & 15);
            float hv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = rt * 64 + 16 * (wave & 3) + 4 * (lane >> 4) + r;
                hv[r] = ld1<true>(hin + (long)(s < B ? s : 0) * H + i);
            }
            WsPanel<H, 64> ra;
            WsPanel<32, H> rw;
            ws_panel_issue(ra, Dl, H, rt * 64, 64, (int)B, 0, tid);
            ws_panel_issue(rw, Wl, H, 0, H, H, ct * 32, tid);
            ws_panel_commit(ra, sA, HS, 64, tid);
            ws_panel_commit(rw, sWc, kWsPS, H, tid);
            __syncthreads();
            const f32x4 acc = ws_mfma_row_col_c<NCH>(sA + 16 * (wave & 3) * HS, HS, sWc + 16 * (wave >> 2), lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = rt * 64 + 16 * (wave & 3) + 4 * (lane >> 4) + r;
                if (s < B) dout[(long)s * H + i] = acc[r] * act_bwd(hv[r], act);
            }
            __syncthreads();
        };

        // WG(l): one tile of dW_l = D_l^T . h_{l-1} over all rows (h_{-1} = the inputs), db_l with input tile 0.
        // Tiles are 64 x 64 when both [Bk][68] panels fit in LDS (B <= 256, H >= 64), else 32 x 32 with a K split.
        const bool wide = H >= 64 && Bk <= 256;
        const int TW = wide ? 64 : 32, PSW = wide ? kWsPSW : kWsPS;
        const int n_ot = H / TW, n_it0 = (in_dim + TW - 1) / TW;
        auto n_wg_jobs = [&](const int l) { return n_ot * (l >= 1 ? n_ot : n_it0); };
        auto wgrad_job = [&](const int l, const int jj) {
            const float* Dl = dbuf + (long)l * plane;
            const int n_it = l >= 1 ? n_ot : n_it0;
            const int ot = jj / n_it, itile = jj - ot * n_it;
            float* sD = sP;                                   // D_l[:, ot*TW .. +TW]      [Bk][PSW]
            float* sX = sP + Bk * PSW;                        // h_{l-1}[:, it*TW .. +TW]  [Bk][PSW]
            WsPanel<32, 512> rd;
            WsPanel<64, 256> rdw;
            if (wide) ws_panel_issue(rdw, Dl, H, 0, Bk, (int)B, ot * 64, tid);
            else ws_panel_issue(rd, Dl, H, 0, Bk, (int)B, ot * 32, tid);
            if (l >= 1) {
                if (wide) {
                    WsPanel<64, 256> rx;
                    ws_panel_issue(rx, hbuf + (long)(l - 1) * plane, H, 0, Bk, (int)B, itile * 64, tid);
                    ws_panel_commit(rx, sX, kWsPSW, Bk, tid);
                } else {
                    WsPanel<32, 512> rx;
                    ws_panel_issue(rx, hbuf + (long)(l - 1) * plane, H, 0, Bk, (int)B, itile * 32, tid);
                    ws_panel_commit(rx, sX, kWsPS, Bk, tid);
                }
            } else {
                // the inputs of the mini-batch were published (gathered, zero padded to 64 columns) by the forward jobs
                if (wide) {
                    WsPanel<64, 256> rx;
                    ws_panel_issue(rx, xbuf, 64, 0, Bk, Bp, 0, tid);
                    ws_panel_commit(rx, sX, kWsPSW, Bk, tid);
                } else {
                    WsPanel<32, 512> rx;
                    ws_panel_issue(rx, xbuf, 64, 0, Bk, Bp, itile * 32, tid);
                    ws_panel_commit(rx, sX, kWsPS, Bk, tid);
                }
            }
            if (wide) ws_panel_commit(rdw, sD, kWsPSW, Bk, tid);
            else ws_panel_commit(rd, sD, kWsPS, Bk, tid);
            __syncthreads();
            const long ldw = l >= 1 ? H : in_dim;
            float* dst_b = itile == 0 ? G + offB(l) : nullptr;
            if (wide)
                sumsq += ws_wgrad_tile_wide(sD, sX, sRed, Bk, G + offW(l), ldw, ot * 64, itile * 64, (int)ldw, dst_b,
                                            u.grad_scale, tid, wave, lane);
            else
                sumsq += ws_wgrad_tile(sD, sX, sRed, Bk, G + offW(l), ldw, ot * 32, itile * 32, (int)ldw, dst_b,
                                       u.grad_scale, tid, wave, lane);
            __syncthreads();
        };

        // OUT: output layer (+ log_std): row-block partials -> gradient, in block order
        auto out_job = [&]() {
            const float sc = u.grad_scale;
            for (long idx = tid; idx < seg_len; idx += kThreadsU) {
                float acc = 0.f;
                for (int g0 = 0; g0 < n_hb; g0 += 8) {
                    float pv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) pv[q] = ld1<true>(outpart + (long)(g0 + q < n_hb ? g0 + q : 0) * seg_len + idx);
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (g0 + q < n_hb) acc += pv[q];
                }
                G[seg_off + idx] = acc;
                sumsq += (double)(acc * sc) * (acc * sc);
            }
        };

        // loss bookkeeping of this network (the partials are complete since the HEAD barrier), Adam step counter +
        // bias corrections: one wave of the last worker
        auto bookkeeping = [&]() {
            if (w == W - 1 && tid < 64) {
                float p0 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f, p7 = 0.f;
                for (int gg = lane; gg < n_hb; gg += 64) {
                    const float* a = u.loss_partials + ((long)which * n_hb + gg) * 8;
                    p0 += ld1<true>(a); p2 += ld1<true>(a + 2); p3 += ld1<true>(a + 3); p4 += ld1<true>(a + 4); p7 += ld1<true>(a + 7);
                }
                p0 = wave_sum(p0); p2 = wave_sum(p2); p3 = wave_sum(p3); p4 = wave_sum(p4); p7 = wave_sum(p7);
                if (lane == 0) {
                    const float n = (float)B;
                    if (which == 0) {
                        const float surr = p0 / n, ent = p3 / n, kl = p4 / n;
                        float total = surr;
                        if (u.entropy_weight != 0.0f) total -= u.entropy_weight * ent;
                        if (u.kl_loss_weight > 0.0f) total += u.kl_loss_weight * kl;
                        u.totals[0] += (double)surr; u.totals[1] += (double)total;
                        u.totals[3] += (double)ent; u.totals[4] += (double)kl;
                        u.totals[5] += (double)ld1<true>(u.loss_partials + 5); u.totals[6] += (double)ld1<true>(u.loss_partials + 6);
                        u.totals[7] += p7 > 0.f ? 1.0 : 0.0;
                        u.totals[8] += 1.0;
                    } else {
                        u.totals[2] += (double)(p2 / n);
                    }
                    const int64_t t = u.step_counts[which] + 1;           // only this lane touches the counter in the launch
                    u.step_counts[which] = t;
                    u.norm_scratch[2 + 2 * which] = 1.0 - pow((double)u.beta1, (double)t);
                    u.norm_scratch[3 + 2 * which] = sqrt(1.0 - pow((double)u.beta2, (double)t));
                }
            }
        };
#define PPOAF_WS_BARRIER(k)                                                           \
        PPOAF_WSTAMP(k);                                                              \
        if (!ws_barrier(ctl, which, w, W, ++epoch, budget, s_ok)) return;             \
        PPOAF_WSTAMP((k) + 1);

        // ================================================================ forward
        const int n_tile = n_rt * nct;
        if (!fuse0) {
            for (int j = w; j < n_tile; j += W) f0_job(j);
            PPOAF_WS_BARRIER(0)
        }
        const int l_fwd_end = tail ? depth - 1 : depth;           // hidden layers [1, l_fwd_end) as tile phases
        for (int l = 1; l < l_fwd_end; ++l) {
            for (int j = w; j < n_tile; j += W) f_job(l, j, fuse0 && l == 1);
            PPOAF_WS_BARRIER(0)
        }
        // ================================================================ HEAD (+ tail)
        for (int g = w; g < n_hb; g += W) head_job(g, tail);
        PPOAF_WS_BARRIER(2)

        // ================================================================ backward
        // D_{d-1} is known (and D_{d-2} after a tail phase).  {DG(m) -> D_{m-1}, WG(m+1)} for m = m0 .. 1, then
        // {WG(1), WG(0), OUT, bookkeeping}: every wgrad runs one phase after the dgrad that produced its D.
        for (int m = tail ? depth - 2 : depth - 1; m >= 1; --m) {
            const int n_w = m + 1 <= depth - 1 ? n_wg_jobs(m + 1) : 0;
            if (n_w > 0 && 2 * n_w <= W) {
                // a wgrad tile costs about two dgrad tiles: the last n_w workers take one wgrad tile each, the dgrad
                // tiles go round the others
                const int n_d = W - n_w;
                if (w >= n_d) wgrad_job(m + 1, w - n_d);
                else for (int j = w; j < n_tile; j += n_d) dgrad_job(m, j);
            } else {
                for (int j = w; j < n_tile + n_w; j += W) {
                    if (j < n_tile) dgrad_job(m, j);
                    else wgrad_job(m + 1, j - n_tile);
                }
            }
            PPOAF_WS_BARRIER(4)
        }
        {
            const int n1 = depth >= 2 ? n_wg_jobs(1) : 0, n0 = n_wg_jobs(0);
            const int n_jobs = n1 + n0 + 1;
            // the bookkeeping worker (W - 1) gets a tile only when there are more jobs than other workers
            for (int j = w; j < n_jobs; j += W) {
                if (j < n1) wgrad_job(1, j);
                else if (j < n1 + n0) wgrad_job(0, j - n1);
                else out_job();
            }
            bookkeeping();
            const double q = block_sum(sumsq, s_red);
            if (tid == 0) ctl->norm_partials[which][w] = q;
            PPOAF_WS_BARRIER(8)
        }
        if (ka->xchg_ranks > 0) {
            // N > 1: the gradient is complete in G.  Every worker sends the columns it owns (the ones it applies Adam
            // to) through the exchange, writes the cross-rank sum back to G and replaces the norm partial by that of
            // the summed gradient -- one phase and one barrier more than a single rank has.
            const long lo4x = nd.offset >> 2, hi4x = (nd.offset + nd.size) >> 2;
            const long i0 = lo4x + (long)w * kThreadsU + tid, cs = (long)W * kThreadsU;
            const long long seq = ++xseq;
            const long slot = (long)(seq & 1) * ka->x.n4;
            for (long i = i0; i < hi4x; i += cs) ws_xchg_store(ka, slot + i, ld4<true>(u.grads + 4 * i));
            ws_xchg_publish_wait(ka, seq, xg);
            double q2 = 0.0;
            for (long i = i0; i < hi4x; i += cs) {
                const float4 acc = ws_xchg_sum(ka, slot + i, ld4<true>(u.grads + 4 * i));
                reinterpret_cast<float4*>(u.grads)[i] = acc;
                q2 += xchg_sq(acc, u.grad_scale);
            }
            q2 = block_sum(q2, s_red);
            if (tid == 0) ctl->norm_partials[which][w] = q2;
            PPOAF_WS_BARRIER(12)
        }

        // ================================================================ clip + Adam on this worker's columns of the
        // network; norm = partials in worker order.  The parameter / moment / gradient loads depend on nothing
        // computed here, so the first kWsAdamCols columns of every thread are requested before the norm is read.
        {
            constexpr int kWsAdamCols = 3;
            const long lo4 = nd.offset >> 2, hi4 = (nd.offset + nd.size) >> 2;
            const long idx0 = lo4 + (long)w * kThreadsU + tid, cstride = (long)W * kThreadsU;
            float4 pp[kWsAdamCols], gg[kWsAdamCols], mm[kWsAdamCols], vv[kWsAdamCols];
#pragma unroll
            for (int c = 0; c < kWsAdamCols; ++c) {
                const long ix = idx0 + c * cstride < hi4 ? idx0 + c * cstride : lo4;
                pp[c] = ld4<true>(u.params + 4 * ix); gg[c] = ld4<true>(u.grads + 4 * ix);
                mm[c] = ld4<true>(u.exp_avg + 4 * ix); vv[c] = ld4<true>(u.exp_avg_sq + 4 * ix);
            }
            if (tid < 64) {
                const double part = lane < W ? ld1<true>(&ctl->norm_partials[which][lane]) : 0.0;
                const double bc1 = ld1<true>(u.norm_scratch + 2 + 2 * which), bc2 = ld1<true>(u.norm_scratch + 3 + 2 * which);
                double sq = 0.0;
                for (int k = 0; k < W; ++k) sq += __shfl(part, k, 64);          // worker order, every lane the same sum
                if (lane == 0) {
                    s_norm[0] = sq;
                    s_step[0] = (float)((double)u.lr[0] / bc1);
                    s_step[1] = (float)bc2;
                }
            }
            __syncthreads();
            const float total_norm = (float)sqrt(s_norm[0]);
            float coef = 1.0f;
            if (u.max_norm > 0.f) coef = fminf(u.max_norm / (total_norm + 1e-6f), 1.0f);
            const float gs = u.grad_scale * coef;
            const float step_size = s_step[0], bc2_sqrt = s_step[1];
            const float beta1 = u.beta1, beta2 = u.beta2, adam_eps = u.adam_eps;
#define PPOAF_ADAM1(c)                                                   \
            {                                                            \
                const float gi = gr.c * gs;                              \
                m.c = beta1 * m.c + (1.0f - beta1) * gi;                 \
                v.c = beta2 * v.c + (1.0f - beta2) * gi * gi;            \
                p.c = p.c - step_size * (m.c / (sqrtf(v.c) / bc2_sqrt + adam_eps)); \
            }
            int c = 0;
            for (long idx = idx0; idx < hi4; idx += cstride, ++c) {
                float4 p, gr, m, v;
                if (c < kWsAdamCols) {
                    // (compile-time indices only: a runtime index would send the arrays to scratch)
                    p = c == 0 ? pp[0] : (c == 1 ? pp[1] : pp[2]); gr = c == 0 ? gg[0] : (c == 1 ? gg[1] : gg[2]);
                    m = c == 0 ? mm[0] : (c == 1 ? mm[1] : mm[2]); v = c == 0 ? vv[0] : (c == 1 ? vv[1] : vv[2]);
                } else {
                    p = ld4<true>(u.params + 4 * idx); gr = ld4<true>(u.grads + 4 * idx);
                    m = ld4<true>(u.exp_avg + 4 * idx); v = ld4<true>(u.exp_avg_sq + 4 * idx);
                }
                PPOAF_ADAM1(x) PPOAF_ADAM1(y) PPOAF_ADAM1(z) PPOAF_ADAM1(w)
                reinterpret_cast<float4*>(const_cast<float*>(u.params))[idx] = p;
                reinterpret_cast<float4*>(u.exp_avg)[idx] = m;
                reinterpret_cast<float4*>(u.exp_avg_sq)[idx] = v;
            }
#undef PPOAF_ADAM1
        }
        PPOAF_WSTAMP(10);
        ws_arrive(ctl, which, w, ++epoch);
        if (it + 1 < n_mb) build_row_table(mb + 1);
        if (!ws_wait(ctl, which, W, epoch, budget, s_ok)) return;
        PPOAF_WSTAMP(11);
#undef PPOAF_WS_BARRIER
    }
    if (ka0->xchg_ranks > 0 && threadIdx.x == 0) {
        ka0->x.group_seq[xg] = xseq;
        if (xg == 0) ka0->x.words[0] = xseq;
    }
    // the cursor moves once BOTH networks are done with the launch (each read it when it started)
    if (w == 0 && threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 1u) ka0->u.cursor[0] = cursor0 + n_mb;
    }
}

// Row-tiled mode of a network's worker group (hidden widths <= 128, where a 16-row workgroup is far from its MFMA
// floor): phase 1 is the three-launch chain's fwd_bwd body on ceil(B/16) of the W workers (slabs), phase 2 folds the
// slabs for the columns this worker owns -- the sums stay in registers, next to the parameter / moment values that
// were requested before the slab loads -- and phase 3 applies clip + Adam to them: three barriers per mini-batch, no
// gradient round trip.  At most kRtCols float4 columns per thread (W * 512 * kRtCols * 4 floats per network).
constexpr int kRtCols = 4;

template <int WHICH, int H>
__device__ __forceinline__ void rt_worker(KWsArgs* ka0, WsCtl* ctl, const int w, const int n_mb, const long long budget,
                                          int* s_ok, double* s_red, double* s_norm, float* s_step) {
    constexpr int which = WHICH;
    typedef const UpdateDev __attribute__((address_space(4))) KU;
    const long cursor0 = ka0->u.cursor[0];
    const int W = ka0->ws.W;
    unsigned epoch = 0;
    const unsigned xg = (unsigned)(which * kWsMaxWorkers + w);                // exchange group of this worker (N > 1)
    long long xseq = ka0->xchg_ranks > 0 ? ka0->x.group_seq[xg] : 0;
    PPOAF_WSTAMP_BEGIN();
    for (int it = 0; it < n_mb; ++it) {
        KWsArgs* ka = ka0;
        asm volatile("" : "+s"(ka));
        KU& u = ka->u;
        int tid_l = threadIdx.x;
        asm volatile("" : "+v"(tid_l));
        __builtin_assume(tid_l >= 0 && tid_l < kThreadsU);
        const int tid = tid_l, lane = tid & 63;
        const int n_wg = u.n_wg;

        // ---- phase 1: forward + backward of 16 rows -> this worker's slab, loss partials.  The wait for the previous
        //      mini-batch's Adam phase (its arrive is at the bottom of the loop) sits inside the body, after the row /
        //      index loads and before the first parameter load.
        {
            const unsigned adam_epoch = epoch;
            auto adam_done = [&]() -> bool { return it == 0 || ws_wait(ctl, which, W, adam_epoch, budget, s_ok); };
            if (w < n_wg) {
                ppo_update_fwd_bwd_body<H / 16, true, KU>(u, which, w, it, adam_done);
                if (it > 0 && *s_ok == 0) return;
            } else if (!adam_done()) {
                return;
            }
        }
        PPOAF_WSTAMP(0);
        if (!ws_barrier(ctl, which, w, W, ++epoch, budget, s_ok)) return;
        PPOAF_WSTAMP(1);

        // ---- phase 2: slabs -> gradient columns (slab order) in registers; bookkeeping; squared-norm partial
        const auto& nd = u.net[which];
        const long n4_all = u.bucket_total >> 2;
        const long lo4 = nd.offset >> 2, hi4 = (nd.offset + nd.size) >> 2;
        const float4* sl = reinterpret_cast<const float4*>(u.slabs);
        float4 p[kRtCols], m[kRtCols], v[kRtCols], gr[kRtCols];
        long idx[kRtCols];
#pragma unroll
        for (int c = 0; c < kRtCols; ++c) {
            idx[c] = lo4 + ((long)c * W + w) * kThreadsU + tid;
            const long ix = idx[c] < hi4 ? idx[c] : lo4;
            p[c] = ld4<true>(u.params + 4 * ix);
            m[c] = ld4<true>(u.exp_avg + 4 * ix);
            v[c] = ld4<true>(u.exp_avg_sq + 4 * ix);
        }
        double sumsq = 0.0;
        const float sc = u.grad_scale;
        const int xr = ka->xchg_ranks;
#pragma unroll
        for (int c = 0; c < kRtCols; ++c) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx[c] < hi4) {                                       // uniform per wave except at the range end
                for (int g0 = 0; g0 < n_wg; g0 += 16) {              // 16 slabs in flight: one L2 round trip at B = 256
                    float4 t[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        t[k] = ld4<true>(reinterpret_cast<const float*>(sl + (long)(g0 + k < n_wg ? g0 + k : 0) * n4_all + idx[c]));
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (g0 + k < n_wg) { acc.x += t[k].x; acc.y += t[k].y; acc.z += t[k].z; acc.w += t[k].w; }
                }
                if (xr == 0) sumsq += xchg_sq(acc, sc);
            }
            gr[c] = acc;
        }
        if (xr > 0) {
            // N > 1: this rank's column sums -> exchange slot, the ranks' sums added in rank order; the clip norm is
            // that of the cross-rank gradient (mpi_avg_gradients before clip_grad_norm_, ppo_policy.py:1035-1038)
            const long long seq = ++xseq;
            const long slot = (long)(seq & 1) * ka->x.n4;
#pragma unroll
            for (int c = 0; c < kRtCols; ++c)
                if (idx[c] < hi4) ws_xchg_store(ka, slot + idx[c], gr[c]);
            ws_xchg_publish_wait(ka, seq, xg);
#pragma unroll
            for (int c = 0; c < kRtCols; ++c)
                if (idx[c] < hi4) {
                    gr[c] = ws_xchg_sum(ka, slot + idx[c], gr[c]);
                    sumsq += xchg_sq(gr[c], sc);
                }
        }
        if (w == W - 1 && tid < 64) {
            const long B = u.B;
            float p0 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f, p7 = 0.f;
            for (int gg = lane; gg < n_wg; gg += 64) {
                const float* a = u.loss_partials + ((long)which * n_wg + gg) * 8;
                p0 += ld1<true>(a); p2 += ld1<true>(a + 2); p3 += ld1<true>(a + 3); p4 += ld1<true>(a + 4); p7 += ld1<true>(a + 7);
            }
            p0 = wave_sum(p0); p2 = wave_sum(p2); p3 = wave_sum(p3); p4 = wave_sum(p4); p7 = wave_sum(p7);
            if (lane == 0) {
                const float n = (float)B;
                if (which == 0) {
                    const float surr = p0 / n, ent = p3 / n, kl = p4 / n;
                    float total = surr;
                    if (u.entropy_weight != 0.0f) total -= u.entropy_weight * ent;
                    if (u.kl_loss_weight > 0.0f) total += u.kl_loss_weight * kl;
                    u.totals[0] += (double)surr; u.totals[1] += (double)total;
                    u.totals[3] += (double)ent; u.totals[4] += (double)kl;
                    u.totals[5] += (double)ld1<true>(u.loss_partials + 5); u.totals[6] += (double)ld1<true>(u.loss_partials + 6);
                    u.totals[7] += p7 > 0.f ? 1.0 : 0.0;
                    u.totals[8] += 1.0;
                } else {
                    u.totals[2] += (double)(p2 / n);
                }
                const int64_t t = u.step_counts[which] + 1;
                u.step_counts[which] = t;
                u.norm_scratch[2 + 2 * which] = 1.0 - pow((double)u.beta1, (double)t);
                u.norm_scratch[3 + 2 * which] = sqrt(1.0 - pow((double)u.beta2, (double)t));
            }
        }
        {
            const double q = block_sum(sumsq, s_red);
            if (tid == 0) ctl->norm_partials[which][w] = q;
        }
        PPOAF_WSTAMP(2);
        if (!ws_barrier(ctl, which, w, W, ++epoch, budget, s_ok)) return;
        PPOAF_WSTAMP(3);

        // ---- phase 3: clip + Adam on the columns held in registers; norm = partials in worker order
        if (tid < 64) {
            const double part = lane < W ? ld1<true>(&ctl->norm_partials[which][lane]) : 0.0;
            const double bc1 = ld1<true>(u.norm_scratch + 2 + 2 * which), bc2 = ld1<true>(u.norm_scratch + 3 + 2 * which);
            double sq = 0.0;
            for (int k = 0; k < W; ++k) sq += __shfl(part, k, 64);
            if (lane == 0) {
                s_norm[0] = sq;
                s_step[0] = (float)((double)u.lr[0] / bc1);
                s_step[1] = (float)bc2;
            }
        }
        __syncthreads();
        {
            const float total_norm = (float)sqrt(s_norm[0]);
            float coef = 1.0f;
            if (u.max_norm > 0.f) coef = fminf(u.max_norm / (total_norm + 1e-6f), 1.0f);
            const float gs = sc * coef;
            const float step_size = s_step[0], bc2_sqrt = s_step[1];
            const float beta1 = u.beta1, beta2 = u.beta2, adam_eps = u.adam_eps;
#pragma unroll
            for (int c = 0; c < kRtCols; ++c) {
                if (idx[c] < hi4) {
#define PPOAF_ADAM1(f)                                                               \
                    {                                                                \
                        const float gi = gr[c].f * gs;                               \
                        m[c].f = beta1 * m[c].f + (1.0f - beta1) * gi;               \
                        v[c].f = beta2 * v[c].f + (1.0f - beta2) * gi * gi;          \
                        p[c].f = p[c].f - step_size * (m[c].f / (sqrtf(v[c].f) / bc2_sqrt + adam_eps)); \
                    }
                    PPOAF_ADAM1(x) PPOAF_ADAM1(y) PPOAF_ADAM1(z) PPOAF_ADAM1(w)
#undef PPOAF_ADAM1
                    reinterpret_cast<float4*>(const_cast<float*>(u.params))[idx[c]] = p[c];
                    reinterpret_cast<float4*>(u.exp_avg)[idx[c]] = m[c];
                    reinterpret_cast<float4*>(u.exp_avg_sq)[idx[c]] = v[c];
                    reinterpret_cast<float4*>(u.grads)[idx[c]] = gr[c];
                }
            }
        }
        PPOAF_WSTAMP(4);
        ws_arrive(ctl, which, w, ++epoch);            // the matching wait is inside the next mini-batch's phase 1
        PPOAF_WSTAMP(5);
    }
    if (ka0->xchg_ranks > 0 && threadIdx.x == 0) {
        ka0->x.group_seq[xg] = xseq;
        if (xg == 0) ka0->x.words[0] = xseq;
    }
    if (w == 0 && threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 1u) ka0->u.cursor[0] = cursor0 + n_mb;
    }
}

template <int HA, int HC, bool LAYERED_A, bool LAYERED_C>
__global__ __launch_bounds__(kThreadsU) void ppo_update_ws_kernel(WsArgs a, WsCtl* ctl, int n_mb, long long budget) {
    __shared__ int s_ticket, s_ok;
    __shared__ double s_red[17];
    __shared__ double s_norm[2];
    __shared__ float s_step[4];
    KWsArgs* ka = (KWsArgs*)__builtin_amdgcn_kernarg_segment_ptr();     // `a` is the first kernel argument
    const int xcc = (int)hw_xcc_id();
    const int which = xcc == a.ws.xcc[0] ? 0 : (xcc == a.ws.xcc[1] ? 1 : -1);
    if (which < 0) return;                                    // uniform per workgroup
    if (threadIdx.x == 0) s_ticket = (int)atomicAdd(&ctl->tickets[which], 1u);
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(s_ticket);
    if (w >= a.ws.W) return;
    if (which == 0) {
        if (LAYERED_A) ws_worker<0, HA>(ka, ctl, w, n_mb, budget, &s_ok, s_red, s_norm, s_step);
        else rt_worker<0, HA>(ka, ctl, w, n_mb, budget, &s_ok, s_red, s_norm, s_step);
    } else {
        if (LAYERED_C) ws_worker<1, HC>(ka, ctl, w, n_mb, budget, &s_ok, s_red, s_norm, s_step);
        else rt_worker<1, HC>(ka, ctl, w, n_mb, budget, &s_ok, s_red, s_norm, s_step);
    }
}

static size_t ws_lds_floats(const NetDev& n, long B) {
    const size_t H = n.H, in_pad = (n.in_dim + 15) & ~15, Bk = (B + 15) & ~15L;
    const size_t K = in_pad > H ? in_pad : H;
    size_t p = 96 * (K + 4) + 64 * 20 + H * 20 + H;                      // forward: A [64][K+4] + W [32][K+4] (+ on-the-fly layer 0)
    const size_t dg = 64 * (H + 4) + H * kWsPS, wg = 2 * Bk * ((H >= 64 && Bk <= 256) ? kWsPSW : kWsPS);
    size_t hd = 16 * (H + 4) + 8 * H + 16;                               // head
    if (H <= 128) hd += 32 * (H + 4) + H + H * (H + 4);                  //   + tail: h_{d-2}, D_{d-1}, b_{d-1}, W_{d-1}
    if (dg > p) p = dg;
    if (wg > p) p = wg;
    if (hd > p) p = hd;
    return 2048 + 16 + 48 + 128 + 256 + 256 + 1536 + p;
}

// which decomposition each network's worker group runs: bit 0 actor, bit 1 critic set = layered (weight-stationary
// tiles), clear = row-tiled (fwd_bwd body + in-register slab fold); mask < 0: automatic -- layered for 256-wide
// networks (a 16-row workgroup is then at its MFMA floor), row-tiled below.
static int ws_resolve_mask(const UpdateDev& u, int mask) {
    if (mask >= 0) return mask & 3;
    return (u.net[0].H >= 256 ? 1 : 0) | (u.net[1].H >= 256 ? 2 : 0);
}

static int ws_check_shapes(const UpdateDev& u, int mask, int workers) {
    mask = ws_resolve_mask(u, mask);
    for (int w = 0; w < 2; ++w) {
        const NetDev& n = u.net[w];
        PPOAF_REQUIRE(n.depth >= 1 && (n.H == 32 || n.H == 64 || n.H == 128 || n.H == 256),
                      "ppo_update_ws: hidden width %d / depth %d not supported (32, 64, 128 or 256; depth >= 1)", n.H, n.depth);
        PPOAF_REQUIRE(n.out_dim >= 1 && n.out_dim <= 8, "ppo_update_ws: out_dim=%d", n.out_dim);
        if (mask & (1 << w)) {
            PPOAF_REQUIRE(n.in_dim >= 1 && n.in_dim <= 64, "ppo_update_ws: in_dim=%d (1..64 in the layered mode)", n.in_dim);
            PPOAF_REQUIRE(ws_lds_floats(n, u.B) * 4 <= 156 * 1024, "ppo_update_ws: needs %zu B of LDS", ws_lds_floats(n, u.B) * 4);
        } else {
            PPOAF_REQUIRE(u.n_wg <= workers, "ppo_update_ws: batch size %ld needs %d row-tile workers, %d given", u.B, u.n_wg, workers);
            PPOAF_REQUIRE(n.size <= (long)workers * kThreadsU * kRtCols * 4,
                          "ppo_update_ws: network of %ld floats exceeds the in-register fold of %d workers", n.size, workers);
            PPOAF_REQUIRE(rowtile_lds_floats(n) * 4 <= 156 * 1024, "ppo_update_ws: needs %zu B of LDS", rowtile_lds_floats(n) * 4);
        }
    }
    const int ha = u.net[0].H, hc = u.net[1].H;
    PPOAF_REQUIRE((ha == hc && (ha == 32 || ha == 64 || ha == 128 || ha == 256)) || (ha == 128 && hc == 256) || (ha == 64 && hc == 128),
                  "ppo_update_ws: hidden widths (actor %d, critic %d) not instantiated", ha, hc);
    const bool same = mask == 0 || mask == 3;
    PPOAF_REQUIRE(same || (ha == 128 && hc == 256 && mask == 2),
                  "ppo_update_ws: mode mask %d not instantiated for widths (%d, %d)", mask, ha, hc);
    PPOAF_REQUIRE(u.B >= 16 && u.B <= 512, "ppo_update_ws: batch size %ld not supported (16..512)", u.B);
    PPOAF_REQUIRE(u.n_rows < (1L << 31), "ppo_update_ws: %ld rows", u.n_rows);
    return PPOAF_OK;
}

template <int HA, int HC, bool LA, bool LC>
static int ws_launch(const WsArgs& a, WsCtl* ctl, int n_mb, long long budget, size_t lds, hipStream_t s, hipEvent_t e0,
                     hipEvent_t e1) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_update_ws_kernel<HA, HC, LA, LC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        attr_set = true;
    }
    // one workgroup per CU of the whole device: those on the two target XCDs that draw a worker ticket stay
    if (e0 || e1)        // the kernel's own begin / end stamped into the events (bench.py: roofline_update)
        hipExtLaunchKernelGGL((ppo_update_ws_kernel<HA, HC, LA, LC>), dim3(256), dim3(kThreadsU), lds, s, e0, e1, 0, a, ctl, n_mb, budget);
    else
        hipLaunchKernelGGL((ppo_update_ws_kernel<HA, HC, LA, LC>), dim3(256), dim3(kThreadsU), lds, s, a, ctl, n_mb, budget);
    return check_launch("ppo_update_ws");
}


// ------------------------------------------------------------------------------------------------------------
// Split-wgrad chain (the default single-rank chain): the launch between fwd_bwd<SPLIT> and Adam.  fwd_bwd's workgroups
// have published the mini-batch's inputs, hidden activations and dLoss/dz as [Bp][H] panels (u.sp); here ONE 4-wave
// workgroup forms one 16 x 16 tile of dW_l = dz_l^T . h_{l-1} over ALL B rows: every lane requests its MFMA operands
// straight from the panels (element (k, o) of dz and (k, i) of h for its k = 4 c + lane / 16: no LDS staging, all loads
// of the tile in flight at once -- the panels were written by other XCDs a moment ago, so the launch is one cold round
// trip long), wave w takes K chunks w, w + 4, ...; the four partial tiles are folded through LDS in wave order; db_l =
// the column sums of dz_l, on the tiles of input tile 0.  One workgroup per network folds the output layer's per-block
// partials in block order; the last one does the per-mini-batch bookkeeping.  Every workgroup leaves a pair of
// squared-norm partials (scaled gradients, double) for the Adam launch to add in workgroup order.  No slabs, no atomics:
// 8.7 MB of slab round trip per mini-batch at C2 become 1.6 MB of panels, and the hidden-layer wgrad MFMAs leave
// fwd_bwd's dependent chain.
// ------------------------------------------------------------------------------------------------------------
// MAXC = chunks of 16 rows a wave may own (B <= 512: 32 chunks over 4 waves)
template <int H>
__device__ __forceinline__ double split_wgrad_job(const UpdateDev& u, const int which, const int job, float* sFold /* [3][2][256] + [4][16] */) {
    constexpr int MAXC = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: the chunk offsets below stay in scalar registers
    const auto& nd = u.net[which];
    const int in_dim = nd.in_dim, depth = nd.depth, out_dim = nd.out_dim;
    const int B = (int)u.B;
    const long plane = (long)u.sp.Bp * H;
    const long szW0 = ((long)H * in_dim + 3) & ~3L;
    auto offW = [&](int l) -> long { return l == 0 ? 0 : szW0 + H + (long)(l - 1) * ((long)H * H + H); };
    auto offB = [&](int l) -> long {
        return l == 0 ? szW0 : offW(l) + (l < depth ? (long)H * H : (((long)out_dim * H + 3) & ~3L));
    };
    float* G = u.grads + nd.offset;
    constexpr int t = H / 16, t2 = (t + 1) / 2;
    const int n_it0 = (in_dim + 15) / 16, p0 = (n_it0 + 1) / 2;
    const int n_hidden = (depth - 1) * t * t2, n_l0 = t * p0;
    const float sc = u.grad_scale;
    double q = 0.0;
    if (job < n_hidden + n_l0) {
        // 16 output rows x 32 input columns: input tiles itile and itile + 1 (the second may not exist: odd tile counts)
        int l, ot, itile, n_it;
        if (job < n_hidden) { l = 1 + job / (t * t2); const int jj = job % (t * t2); ot = jj / t2; itile = 2 * (jj % t2); n_it = t; }
        else { l = 0; const int jj = job - n_hidden; ot = jj / p0; itile = 2 * (jj % p0); n_it = n_it0; }
        const bool two = itile + 1 < n_it;                    // uniform per workgroup
        // buffer loads: resource = the layer's panel, scalar offset = chunk + row quad, vector offset = the lane's constant
        // byte offset (no vector address arithmetic per load).  Rows of the last chunk beyond B are dead rows of their
        // tile: their dz is zero and their activations finite, exactly as the slab form sums them.
        const long ldx = l >= 1 ? H : 64;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(u.sp.dbuf[which] + (long)l * plane, 0, 0xFFFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            l >= 1 ? u.sp.hbuf[which] + (long)(l - 1) * plane : u.sp.xbuf[which], 0, 0xFFFFFFFF, 0x00020000);
        const unsigned dl = 4u * (unsigned)((lane >> 4) * H + ot * 16 + (lane & 15));
        const unsigned xl = 4u * (unsigned)((lane >> 4) * (int)ldx + itile * 16 + (lane & 15));
        const int nc = (B + 15) >> 4;                         // 16-row chunks of the mini-batch
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
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
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
        // fold the four waves' partial tiles in wave order (waves 1..3 park theirs in LDS)
        if (wave > 0) {
            *reinterpret_cast<f32x4*>(sFold + (((wave - 1) * 2 + 0) * 64 + lane) * 4) = acc0;
            *reinterpret_cast<f32x4*>(sFold + (((wave - 1) * 2 + 1) * 64 + lane) * 4) = acc1;
        }
        // bias: column o = lane & 15 summed over this lane group's rows, then over the 4 lane groups, then over the waves
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 16) sFold[1536 + wave * 16 + lane] = bsum;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                acc0 += *reinterpret_cast<const f32x4*>(sFold + ((w * 2 + 0) * 64 + lane) * 4);
                acc1 += *reinterpret_cast<const f32x4*>(sFold + ((w * 2 + 1) * 64 + lane) * 4);
            }
        }
        const long ldw = l >= 1 ? H : in_dim;
        const int i = itile * 16 + (lane & 15);               // C layout: column = lane & 15, rows 4 (lane >> 4) + r
        float bg = 0.f;
        if (wave == 0 && itile == 0 && lane < 16)
            bg = sFold[1536 + lane] + sFold[1536 + 16 + lane] + sFold[1536 + 32 + lane] + sFold[1536 + 48 + lane];
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ot * 16 + 4 * (lane >> 4) + r;
                if (i < ldw) { G[offW(l) + (long)o * ldw + i] = acc0[r]; q += (double)(acc0[r] * sc) * (acc0[r] * sc); }
                if (two && i + 16 < ldw) { G[offW(l) + (long)o * ldw + i + 16] = acc1[r]; q += (double)(acc1[r] * sc) * (acc1[r] * sc); }
            }
            if (itile == 0 && lane < 16) {
                G[offB(l) + ot * 16 + lane] = bg;
                q += (double)(bg * sc) * (bg * sc);
            }
        }
    } else {
        // output layer (+ log_std): row-block partials -> gradient, in block order
        const long seg_off = offW(depth), seg_len = nd.size - seg_off;
        const float* outpart = u.sp.outpart[which];
        const int n_hb = (B + 15) >> 4;
        for (long idx = tid; idx < seg_len; idx += kWgradThreads) {
            float acc = 0.f;
            for (int g0 = 0; g0 < n_hb; g0 += 8) {
                float pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] = outpart[(long)(g0 + k < n_hb ? g0 + k : 0) * seg_len + idx];
#pragma unroll
                for (int k = 0; k < 8; ++k) if (g0 + k < n_hb) acc += pv[k];
            }
            G[seg_off + idx] = acc;
            q += (double)(acc * sc) * (acc * sc);
        }
    }
    return q;
}

template <int HA, int HC>
__global__ __launch_bounds__(kWgradThreads) void ppo_update_wgrad_kernel(UpdateDev u, int jobs_a, int jobs_c, int per_xcd) {
    __shared__ double s_red[17];
    __shared__ __attribute__((aligned(16))) float s_fold[6 * 256 + 64];
    const int b = blockIdx.x;
    if (b == 8 * per_xcd) { ppo_update_bookkeeping_split(u); return; }           // uniform per workgroup
    const int job = (b & 7) * per_xcd + (b >> 3);             // XCD b % 8 works on one run of the layer-major job list
    double q = 0.0;
    if (job < jobs_a) q = split_wgrad_job<HA>(u, 0, job, s_fold);
    else if (job < jobs_a + jobs_c) q = split_wgrad_job<HC>(u, 1, job - jobs_a, s_fold);
    const bool actor = job < jobs_a;
    q = block_sum(q, s_red);
    if (threadIdx.x == 0) {
        u.norm_scratch[6 + 2 * b] = actor ? q : 0.0;
        u.norm_scratch[7 + 2 * b] = actor ? 0.0 : q;
    }
}

template <int HA, int HC>
static int wgrad_launch(const UpdateDev& u, hipStream_t s) {
    const int ja = split_wgrad_jobs(u.net[0]), jc = split_wgrad_jobs(u.net[1]), px = split_wgrad_per_xcd(u);
    hipLaunchKernelGGL((ppo_update_wgrad_kernel<HA, HC>), dim3((unsigned)(8 * px + 1)), dim3(kWgradThreads), 0, s, u, ja, jc, px);
    return check_launch("ppo_update_wgrad");
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_ppo_update_ws_ctl_bytes(void) { return (int)sizeof(WsCtl); }

extern "C" int ppoaf_ppo_update_ws_workspace_bytes(const ppoaf_ppo_update_args_t* args, int32_t layered_mask, int64_t* bytes_out) {
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    rc = ws_check_shapes(u, layered_mask, kWsMaxWorkers);
    if (rc) return rc;
    PPOAF_REQUIRE(bytes_out, "ppo_update_ws_workspace_bytes: null output");
    *bytes_out = (int64_t)ws_layout(u, nullptr, nullptr);
    return PPOAF_OK;
}

static int ws_update(const ppoaf_ppo_update_args_t* args, int64_t n_minibatches, void* ctl, void* workspace,
                     int64_t workspace_bytes, int32_t workers, int32_t xcc_actor, int32_t xcc_critic,
                     int32_t layered_mask, double wait_seconds, ppoaf_peer_exchange_t* xchg, double xchg_wait_seconds,
                     int32_t xchg_fences, void* start_event, void* stop_event, ppoaf_stream_t stream) {
    hipEvent_t e0 = (hipEvent_t)start_event, e1 = (hipEvent_t)stop_event;
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    PPOAF_REQUIRE(workers >= 1 && workers <= kWsMaxWorkers, "ppo_update_ws: workers=%d (1..%d: one XCD holds 32 CUs)", workers, kWsMaxWorkers);
    rc = ws_check_shapes(u, layered_mask, workers);
    if (rc) return rc;
    const int mask = ws_resolve_mask(u, layered_mask);
    PPOAF_REQUIRE(ctl && (((uintptr_t)ctl) & 15) == 0, "ppo_update_ws: control block missing or misaligned");
    PPOAF_REQUIRE(workspace && (((uintptr_t)workspace) & 255) == 0, "ppo_update_ws: workspace missing or not 256-byte aligned");
    PPOAF_REQUIRE(n_minibatches >= 1 && n_minibatches <= (1 << 20), "ppo_update_ws: n_minibatches=%ld", (long)n_minibatches);
    PPOAF_REQUIRE(xcc_actor >= 0 && xcc_actor < 8 && xcc_critic >= 0 && xcc_critic < 8 && xcc_actor != xcc_critic,
                  "ppo_update_ws: XCDs %d / %d (two different ones of 0..7)", xcc_actor, xcc_critic);
    PPOAF_REQUIRE(wait_seconds > 0.0 && wait_seconds <= 600.0, "ppo_update_ws: wait_seconds=%g", wait_seconds);
    PPOAF_REQUIRE(u.mb_offset == 0, "ppo_update_ws: mb_offset must be 0 (the launch walks the cursor itself)");
    WsDev ws;
    const size_t need = ws_layout(u, &ws, reinterpret_cast<char*>(workspace));
    PPOAF_REQUIRE((size_t)workspace_bytes >= need, "ppo_update_ws: workspace of %ld B, %zu needed", (long)workspace_bytes, need);
    ws.W = workers; ws.xcc[0] = xcc_actor; ws.xcc[1] = xcc_critic;
    size_t lds = 0;
    for (int w = 0; w < 2; ++w) {
        const size_t f = (mask & (1 << w)) ? ws_lds_floats(u.net[w], u.B) : rowtile_lds_floats(u.net[w]);
        if (f > lds) lds = f;
    }
    lds = (lds * 4 + 15) / 16 * 16;
    if (lds < 96 * 1024) lds = 96 * 1024;      // one workgroup per CU: the 256 launched workgroups then cover every CU
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ctl, 0, sizeof(WsCtl), s);            // tickets, error word, barrier epochs
    if (e != hipSuccess) { set_error("ppo_update_ws: memset: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
    const long long budget = (long long)(wait_seconds * 1.0e8);          // wall_clock64 ticks at 100 MHz
    WsArgs a;
    a.u = u; a.ws = ws;
    a.xchg_ranks = 0; a.xchg_fences = 0; a.xchg_ticks = 0;
    a.x = XchgDev();
    if (xchg) {
        PPOAF_REQUIRE(xchg->connected, "ppo_update_ws_exchange: the exchange is not connected");
        PPOAF_REQUIRE(xchg->dev.n4 == (u.bucket_total >> 2), "ppo_update_ws_exchange: exchange made for %ld float4, bucket has %ld",
                      xchg->dev.n4, (long)(u.bucket_total >> 2));
        PPOAF_REQUIRE(xchg->dev.n_ranks == u.n_ranks || !u.normalize_values, "ppo_update_ws_exchange: %d ranks in the exchange, %d in args",
                      xchg->dev.n_ranks, u.n_ranks);
        PPOAF_REQUIRE(xchg_wait_seconds > 0.0 && xchg_wait_seconds <= 600.0, "ppo_update_ws_exchange: xchg_wait_seconds=%g", xchg_wait_seconds);
        static_assert(2 * kWsMaxWorkers <= kXchgMaxGrid, "one exchange group per worker");
        a.x = xchg->dev;
        a.xchg_ranks = xchg->dev.n_ranks;
        // coarse-grained slots are only coherent through the fences; the other kinds may ask for them too
        a.xchg_fences = (xchg_fences || xchg->memory_kind == 3) ? 1 : 0;
        a.xchg_ticks = (long long)(xchg_wait_seconds * 1.0e8);
    }
    WsCtl* c = reinterpret_cast<WsCtl*>(ctl);
    const int ha = u.net[0].H, hc = u.net[1].H, n = (int)n_minibatches;
#define PPOAF_WS_PAIR(A, C)                                                                     \
    if (ha == A && hc == C) {                                                                   \
        if (mask == 3) return ws_launch<A, C, true, true>(a, c, n, budget, lds, s, e0, e1);     \
        if (mask == 0) return ws_launch<A, C, false, false>(a, c, n, budget, lds, s, e0, e1);   \
    }
    PPOAF_WS_PAIR(32, 32) PPOAF_WS_PAIR(64, 64) PPOAF_WS_PAIR(128, 128) PPOAF_WS_PAIR(256, 256)
    PPOAF_WS_PAIR(128, 256) PPOAF_WS_PAIR(64, 128)
#undef PPOAF_WS_PAIR
    if (ha == 128 && hc == 256 && mask == 2) return ws_launch<128, 256, false, true>(a, c, n, budget, lds, s, e0, e1);
    set_error("ppo_update_ws: hidden widths (actor %d, critic %d) not instantiated", ha, hc);
    return PPOAF_E_INVALID;
}

extern "C" int ppoaf_ppo_update_ws(const ppoaf_ppo_update_args_t* args, int64_t n_minibatches, void* ctl, void* workspace,
                                   int64_t workspace_bytes, int32_t workers, int32_t xcc_actor, int32_t xcc_critic,
                                   int32_t layered_mask, double wait_seconds, void* start_event, void* stop_event,
                                   ppoaf_stream_t stream) {
    return ws_update(args, n_minibatches, ctl, workspace, workspace_bytes, workers, xcc_actor, xcc_critic, layered_mask,
                     wait_seconds, nullptr, 0.0, 0, start_event, stop_event, stream);
}

extern "C" int ppoaf_ppo_update_ws_exchange(const ppoaf_ppo_update_args_t* args, int64_t n_minibatches, void* ctl, void* workspace,
                                            int64_t workspace_bytes, int32_t workers, int32_t xcc_actor, int32_t xcc_critic,
                                            int32_t layered_mask, double wait_seconds, ppoaf_peer_exchange_t* xchg,
                                            double xchg_wait_seconds, int32_t xchg_fences, void* start_event,
                                            void* stop_event, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(xchg, "ppo_update_ws_exchange: null exchange (single rank: ppoaf_ppo_update_ws)");
    return ws_update(args, n_minibatches, ctl, workspace, workspace_bytes, workers, xcc_actor, xcc_critic, layered_mask,
                     wait_seconds, xchg, xchg_wait_seconds, xchg_fences, start_event, stop_event, stream);
}

extern "C" int ppoaf_ppo_update_split_workspace_bytes(const ppoaf_ppo_update_args_t* args, int64_t* bytes_out) {
    PPOAF_REQUIRE(args && bytes_out, "ppo_update_split_workspace_bytes: null argument");
    ppoaf_ppo_update_args_t a = *args;
    a.split_workspace = nullptr;
    UpdateDev u;
    int rc = make_update_dev(&a, u);
    if (rc) return rc;
    PPOAF_REQUIRE(u.net[0].in_dim <= 64 && u.net[1].in_dim <= 64 && u.B <= 512,
                  "ppo_update_split_workspace_bytes: the split-wgrad chain covers in_dim <= 64 and B <= 512 (got %d / %d, %ld)",
                  u.net[0].in_dim, u.net[1].in_dim, u.B);
    size_t need = ws_layout(u, nullptr, nullptr);
    if (a.row_pairs) need = pair_region_offset(u) + pair_region_layout(u, nullptr, nullptr);
    *bytes_out = (int64_t)need;
    return PPOAF_OK;
}

extern "C" int ppoaf_ppo_update_row_pairs_error_offset(const ppoaf_ppo_update_args_t* args, int64_t* offset_out) {
    PPOAF_REQUIRE(args && offset_out, "ppo_update_row_pairs_error_offset: null argument");
    ppoaf_ppo_update_args_t a = *args;
    a.split_workspace = nullptr;
    UpdateDev u;
    int rc = make_update_dev(&a, u);
    if (rc) return rc;
    const int ha = u.net[0].H, hc = u.net[1].H;
    const bool pairs = a.row_pairs && ((ha == 128 && hc == 256 && pair_eligible(u.net[1])) ||
                                       (ha == 256 && hc == 256 && pair_eligible(u.net[0]) && pair_eligible(u.net[1])));
    *offset_out = pairs ? (int64_t)pair_region_offset(u) : -1;
    return PPOAF_OK;
}

extern "C" int ppoaf_ppo_update_split_blocks(const ppoaf_ppo_update_args_t* args) {
    UpdateDev u;
    ppoaf_ppo_update_args_t a;
    if (!args) return -1;
    a = *args;
    a.split_workspace = nullptr;
    if (make_update_dev(&a, u)) return -1;
    return split_wgrad_blocks(u);
}

extern "C" int ppoaf_ppo_update_wgrad(const ppoaf_ppo_update_args_t* args, ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    PPOAF_REQUIRE(u.split, "ppo_update_wgrad: args->split_workspace is not set (the slab chain uses ppoaf_ppo_update_reduce)");
    hipStream_t s = (hipStream_t)stream;
    const int ha = u.net[0].H, hc = u.net[1].H;
    if (ha == 32 && hc == 32) return wgrad_launch<32, 32>(u, s);
    if (ha == 64 && hc == 64) return wgrad_launch<64, 64>(u, s);
    if (ha == 128 && hc == 128) return wgrad_launch<128, 128>(u, s);
    if (ha == 256 && hc == 256) return wgrad_launch<256, 256>(u, s);
    if (ha == 128 && hc == 256) return wgrad_launch<128, 256>(u, s);
    if (ha == 64 && hc == 128) return wgrad_launch<64, 128>(u, s);
    set_error("ppo_update_wgrad: hidden widths (actor %d, critic %d) not instantiated", ha, hc);
    return PPOAF_E_INVALID;
}

