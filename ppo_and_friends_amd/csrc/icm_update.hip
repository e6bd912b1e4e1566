// K14: fused ICM mini-batch update (see include/ppoaf_hip.h for the contract).
// One iteration of PPO._icm_batch_train (ppo.py:2487-2567) =
//   icm_encoder_fwd_kernel   obs / next_obs rows -> encoder (4 layers) -> activations in scratch
//   icm_heads_kernel         inverse model | forward model: forward, loss, backward -> slabs,
//                            gradients of the two encodings -> scratch
//   icm_encoder_bwd_kernel   encoder backward for both observations -> slabs
//   icm_reduce_kernel        slabs -> gradient bucket [-> Adam], loss -> totals, cursor++
//
// Work decomposition: as K12 -- 16 rows per workgroup (one MFMA M tile), 8 waves each owning 16
// output columns of an H-wide layer, activations in LDS, weights streamed from L2.  The four
// sub-networks depend on each other across rows' workgroups only through the encodings, so the
// dependency is carried by kernel boundaries (~2 us each on gfx950) rather than grid barriers,
// and each phase runs on 2 * B/16 workgroups: (obs, next_obs) for the encoder phases,
// (inverse, forward model) for the heads.  float32 MFMA (v_mfma_f32_16x16x4_f32) keeps fmaf
// chains exact, weight gradients are written to private slabs and summed in a fixed order.
#include "ppo_update_rowpair.hpp"
#include <algorithm>
#include <cstdlib>

namespace ppoaf {

struct IcmDev {
    int O, H, A, Ain, d_inv, d_fwd, act, discrete;
    long enc_off, inv_off, fwd_off, enc_size, total;
    const float* params; float* grads; float* exp_avg; float* exp_avg_sq; float* slabs;
    int64_t* step_count; const float* lr;
    float beta1, beta2, adam_eps, grad_scale;
    const float* obs; const float* next_obs; const void* actions;
    const int64_t* perm; const int32_t* row_map; long n_rows;
    int64_t* cursor; long B, batch_stride, Bpad;
    float icm_beta; int fused_adam, pregathered;
    float* actE; float* dEnc; float* loss_partials; double* totals;
    int nT, confine;
    // split-wgrad chain (args->split_workspace): the three fwd_bwd kernels form NO weight gradient; they publish every
    // layer's dLoss/dz (and the inputs that are not in scratch already) as [rows][width] panels and the reduce launch
    // becomes icm_wgrad_kernel.  Panels (plane = Bpad * H floats):
    int split;
    float* xO;      // [2][Bpad][XO]   gathered observation rows of the two streams, zero padded to XO = 16 ceil(O / 16)
    float* dE;      // [2][4][plane]   encoder dz, stream-major
    float* hI;      // [d_inv][plane]  inverse model hidden activations     dI: [d_inv][plane] its dz
    float* dI;
    float* oI;      // [Bpad][16]      d(inverse model output), zero padded
    float* hF;      // [d_fwd][plane]  forward model hidden activations     dF: [d_fwd + 1][plane] its dz (last: the output layer)
    float* dF;
    float* aF;      // [Bpad][16]      the forward model's action columns (one-hot / action values), zero padded
    int XO;
};

// one [n_o x n_i] block of some weight matrix = D^T X over all rows (and both observation streams for the encoder)
struct IcmBlk {
    const float* D; const float* X;     // [rows][ldd] / [rows][ldx] panels; segment s adds s * seg_d / s * seg_x floats
    long seg_d, seg_x, w, b;            // w: bucket offset of the block's first weight; b: of its bias (-1: none from this block)
    int n_seg, ldd, ldx, n_o, n_i, ldw, job0, n_ip;
};
constexpr int kIcmMaxBlk = 16;
struct IcmWg { IcmBlk blk[kIcmMaxBlk]; int n_blk, n_jobs; int xcd_job0[9]; };   // XCD x works on jobs [xcd_job0[x], xcd_job0[x + 1])

// 16 rows x H floats of LDS (row stride HS) -> rows [16 g, +16) of a [Bpad][H] panel
template <int H>
__device__ __forceinline__ void icm_publish(const float* __restrict__ src, int HS, float* __restrict__ panel, int g, int tid) {
    float* dst = panel + (long)g * kRows * H;
    for (int i = tid; i < kRows * (H / 4); i += kThreadsU) {
        const int r = i / (H / 4), c4 = i - r * (H / 4);
        *reinterpret_cast<float4*>(dst + (long)r * H + 4 * c4) = *reinterpret_cast<const float4*>(src + r * HS + 4 * c4);
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char icm_smem[];

// args->xcd_half = 1 / 2: the fwd_bwd kernels' workgroups on XCDs 0-3 / 4-7 only (workgroup b is dispatched to XCD b % 8;
// the launch is twice as wide, the other half's workgroups return at once) -> the block index the kernel works on, or -1
__device__ __forceinline__ int icm_block(const IcmDev& u) {
    const int b = blockIdx.x;
    if (!u.confine) return b;
    const int x = b & 7;
    return (x >> 2) != u.confine - 1 ? -1 : ((b >> 3) << 2) | (x & 3);
}

__device__ __forceinline__ void icm_rows(const IcmDev& u, int g, int tid, int* sRow) {
    if (tid < kRows) {
        const long s = (long)g * kRows + tid;
        int row = -1;
        if (s < u.B) {
            if (u.pregathered) {
                row = (int)(u.cursor[0] * u.batch_stride + s);   // tables in shuffled order: no index chain
            } else if (u.perm) {
                const long p = u.perm[u.cursor[0] * u.batch_stride + s];
                if (p >= 0 && p < u.n_rows) row = u.row_map ? u.row_map[p] : (int)p;
            } else {
                row = (int)s;                         // rollout-time reward: the batch is the env batch itself
            }
        }
        sRow[tid] = row;
    }
}

// ------------------------------------------------------------------------------------------------
// encoder forward: blockIdx.x = 2 * g + which (0: obs, 1: next_obs)
// ------------------------------------------------------------------------------------------------
// FUSED (icm_fused_kernel below): the same body as the first phase of a workgroup that goes on to run a model and the
// encoder's backward pass; its activations stay in LDS (they are stored for the wgrad launch all the same).
template <int HT, bool FUSED>
__device__ __forceinline__ void icm_encoder_fwd_body(const IcmDev& u, const int which, const int g, float* smem, float* sLines) {
    constexpr int H = 16 * HT, HS = H + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vb = 2 * g + which;
    const int O = u.O, NT0 = (O + 15) >> 4, INP = 16 * NT0 + 4;
    const float* P = u.params + u.enc_off;
    auto encW = [&](int l) -> long { return l == 0 ? 0 : (long)H * O + H + (long)(l - 1) * (H * H + H); };
    auto encB = [&](int l) -> long { return encW(l) + (l == 0 ? (long)H * O : (long)H * H); };
    int* sRow = reinterpret_cast<int*>(smem);
    float* sX = smem + 16;                      // [16, INP]
    float* sH = sX + kRows * INP;               // 4 x [16, HS]
    float* sScr = sLines + wave * 2 * kLineSlot;                    // this wave's two line slots (mfma_rows_x_lines)
    // every hidden layer's weight set is requested a phase ahead (the first one now): a set asked for when it is needed
    // costs the phase a memory round trip, 14 of them per mini-batch in this chain's three kernels
    const bool has_tile = wave < HT;
    float4 cur[HT], nxt[HT];
    if (has_tile) load_fwd_lines_ld<HT>(P + encW(1), H, wave * 16, lane, cur);
    if (vb == 0 && tid == 0 && u.fused_adam) {
        // Adam step counter and the two bias-correction constants of this mini-batch, computed once (in
        // double, as torch.optim.Adam does) and parked behind the loss partials for the reduce kernel
        const int64_t t = u.step_count[0] + 1;
        u.step_count[0] = t;
        u.loss_partials[2 * u.nT] = (float)((double)u.lr[0] / (1.0 - pow((double)u.beta1, (double)t)));
        u.loss_partials[2 * u.nT + 1] = (float)sqrt(1.0 - pow((double)u.beta2, (double)t));
    }
    icm_rows(u, g, tid, sRow);
    for (int i = tid; i < kRows * INP; i += kThreadsU) sX[i] = 0.f;
    __syncthreads();
    {
        const float* src = which == 0 ? u.obs : u.next_obs;
        for (int idx = tid; idx < kRows * O; idx += kThreadsU) {
            const int s = idx / O, i = idx - s * O;
            const int row = sRow[s];
            if (row >= 0) sX[s * INP + i] = src[(long)row * O + i];
        }
    }
    __syncthreads();
    if (u.split) {          // the layer-0 wgrad's K-panel: this tile's rows, zero padded to XO columns
        float* xo = u.xO + ((long)which * u.Bpad + (long)g * kRows) * u.XO;
        for (int i = tid; i < kRows * u.XO; i += kThreadsU) { const int r = i / u.XO, c = i - r * u.XO; xo[i] = sX[r * INP + c]; }
    }
    // layer 0: K = O padded to 16 (sX zero padded), weights read with a bound check
    for (int nt = wave; nt < HT; nt += kNW) {
        const int o = nt * 16 + (lane & 15);
        const float bv = P[encB(0) + o];
        f32x4 acc = {bv, bv, bv, bv};
        const float* w = P + encW(0) + (long)o * O;
        const float* arow = sX + (lane & 15) * INP;
        for (int k0 = 0; k0 < O; k0 += 16) {
            float bq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 4 * j + (lane >> 4);
                bq[j] = k < O ? w[k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k0 + 4 * j + (lane >> 4)], bq[j], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sH[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], u.act);
    }
    __syncthreads();
    const int o_t = wave * 16 + (lane & 15);
    if (has_tile) {
        const float bv = P[encB(1) + o_t];
        load_fwd_lines_ld<HT>(P + encW(2), H, wave * 16, lane, nxt);
        fwd_tile_store<HT>(mfma_rows_x_lines<HT>(sH, HS, lane, cur, bv, sScr), sH + kRows * HS, u.act, wave, lane);
    }
    __syncthreads();
    if (has_tile) {
        const float bv = P[encB(2) + o_t];
        load_fwd_lines_ld<HT>(P + encW(3), H, wave * 16, lane, cur);
        fwd_tile_store<HT>(mfma_rows_x_lines<HT>(sH + kRows * HS, HS, lane, nxt, bv, sScr), sH + 2L * kRows * HS, u.act, wave, lane);
    }
    __syncthreads();
    if (has_tile) {
        const float bv = P[encB(3) + o_t];
        fwd_tile_store<HT>(mfma_rows_x_lines<HT>(sH + 2L * kRows * HS, HS, lane, cur, bv, sScr), sH + 3L * kRows * HS, -1, wave, lane);
    }
    __syncthreads();
    // activations of all four layers -> scratch [which][l][row][H]
    for (int idx = tid; idx < 4 * kRows * (H / 4); idx += kThreadsU) {
        const int l = idx / (kRows * (H / 4)), rem = idx - l * (kRows * (H / 4));
        const int s = rem / (H / 4), c4 = rem - s * (H / 4);
        const float4 v = *reinterpret_cast<const float4*>(sH + ((long)l * kRows + s) * HS + 4 * c4);
        *reinterpret_cast<float4*>(u.actE + (((long)(which * 4 + l) * u.Bpad) + (long)g * kRows + s) * H + 4 * c4) = v;
    }
}

template <int HT>
__global__ __launch_bounds__(kThreadsU) void icm_encoder_fwd_kernel(IcmDev u) {
    constexpr int HS = 16 * HT + 4;
    const int vb = icm_block(u);
    if (vb < 0 || vb >= 2 * u.nT) return;
    float* smem = reinterpret_cast<float*>(icm_smem);
    const int INP = 16 * ((u.O + 15) >> 4) + 4;
    icm_encoder_fwd_body<HT, false>(u, vb & 1, vb >> 1, smem, smem + 16 + kRows * INP + 4L * kRows * HS);
}

// ------------------------------------------------------------------------------------------------
// heads: blockIdx.x = 2 * g + which (0: inverse model, 1: forward model)
// ------------------------------------------------------------------------------------------------
constexpr int kXS = 20;      // row stride of the padded action tile [16, 16 + 4]

// One launch per mini-batch for the three kernels of this file (icm_fused_kernel): what a workgroup hands to its partner and
// where it finds its own encoding.  Exchanges are rounds of data-tagged records (ppo_update_rowpair.hpp: pair_send / pair_recv).
struct IcmFuse {
    const float* enc_own;      // LDS: this workgroup's stream's encoding (the encoder phase's last plane)
    float* recv;               // LDS [16][HS]: where the partner's encoding gradient lands
    unsigned char* base;       // record region: header (word 0: error), then [phase][tile][half] blocks
    unsigned tag;              // mini-batch index + 1
    __device__ __forceinline__ unsigned char* block(int phase, int g, int half) const {
        return base + kPairHeaderBytes + (((long)phase * kPairMaxTiles + g) * 2 + half) * (long)kPairRecBytes;
    }
};

template <int HT, bool FUSED>
__device__ __forceinline__ void icm_heads_body(const IcmDev& u, const int which, const int g, float* smem, float* sLines, const IcmFuse& fx) {
    constexpr int H = 16 * HT, HS = H + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int act = u.act, A = u.A, Ain = u.Ain;
    const int depth = which == 0 ? u.d_inv : u.d_fwd;
    const long B = u.B;
    int* sRow = reinterpret_cast<int*>(smem);                 // [16]
    float* sAct = smem + 16;                                  // [16][8] actions (float, or int bits)
    float* sBout = smem + 144;                                // [16]
    float* sWout = smem + 160;                                // [8, H]
    float* sXa = sWout + 8 * H;                               // [16, kXS]
    float* sE1 = sXa + kRows * kXS;                           // [16, HS]
    float* sE2 = sE1 + kRows * HS;
    float* sH = sE2 + kRows * HS;                             // depth x [16, HS]
    float* sD0 = sH + (long)depth * kRows * HS;
    float* sD1 = sD0 + kRows * HS;
    float* sOut = sD1 + kRows * HS;                           // [16, 16]
    float* sDOut = sOut + kRows * kMaxOut;                    // [16, 16]
    float* sScr = sLines + wave * 2 * kLineSlot;               // this wave's two line slots (mfma_rows_x_lines)
    __shared__ float red[17];
    // Every weight set is requested a phase (or more) ahead of the MFMAs that consume it -- the first ones here, before
    // anything else; forward sets as whole lines where the rows are whole lines, dgrad sets as buffer loads.
    const bool has_tile = wave < HT;
    const int n0 = wave * 16;
    float4 sa[HT], sb[HT], sc[HT];
    if (has_tile) {
        if (which == 0) {
            load_fwd_lines_ld<HT>(u.params + u.inv_off, 2 * H, n0, lane, sa);          // layer 0, K half of enc_1
            load_fwd_lines_ld<HT>(u.params + u.inv_off + H, 2 * H, n0, lane, sb);      //          K half of enc_2
        } else {
            load_fwd_frags_ld<HT, false>(u.params + u.fwd_off, H + Ain, n0, lane, sa); // layer 0 (rows of H + Ain floats: not lines)
        }
    }

    icm_rows(u, g, tid, sRow);
    for (int i = tid; i < kRows * kXS; i += kThreadsU) sXa[i] = 0.f;
    if constexpr (FUSED) {
        // the two encodings of this tile: mine from the encoder phase (LDS), the other stream's from the partner
        if constexpr (H == 128) pair_send(fx.block(0, g, which), fx.enc_own, HS, 0, fx.tag, tid);
        for (int idx = tid; idx < kRows * (H / 4); idx += kThreadsU) {
            const int s = idx / (H / 4), c4 = idx - s * (H / 4);
            *reinterpret_cast<float4*>((which == 0 ? sE1 : sE2) + s * HS + 4 * c4) = *reinterpret_cast<const float4*>(fx.enc_own + s * HS + 4 * c4);
        }
        if constexpr (H == 128) pair_recv(fx.block(0, g, which ^ 1), which == 0 ? sE2 : sE1, HS, 0, fx.tag, tid, reinterpret_cast<unsigned*>(fx.base));
    } else {
    // encodings of this tile (layer 3 of the encoder scratch)
    for (int idx = tid; idx < 2 * kRows * (H / 4); idx += kThreadsU) {
        const int e = idx / (kRows * (H / 4)), rem = idx - e * (kRows * (H / 4));
        const int s = rem / (H / 4), c4 = rem - s * (H / 4);
        const float4 v = *reinterpret_cast<const float4*>(
            u.actE + (((long)(e * 4 + 3) * u.Bpad) + (long)g * kRows + s) * H + 4 * c4);
        *reinterpret_cast<float4*>((e == 0 ? sE1 : sE2) + s * HS + 4 * c4) = v;
    }
    }
    __syncthreads();
    if (tid < kRows) {
        const int row = sRow[tid];
        if (u.discrete) {
            int a = row >= 0 ? (int)reinterpret_cast<const int64_t*>(u.actions)[row] : 0;
            a = a < 0 ? 0 : (a >= A ? A - 1 : a);
            reinterpret_cast<int*>(sAct)[tid * 8] = a;
            if (row >= 0) sXa[tid * kXS + a] = 1.0f;                              // one-hot (icm.py:198-204)
        } else {
            for (int d = 0; d < A; ++d) {
                const float v = row >= 0 ? reinterpret_cast<const float*>(u.actions)[(long)row * A + d] : 0.f;
                sAct[tid * 8 + d] = v;
                sXa[tid * kXS + d] = v;
            }
        }
    }

    const long sidx = g;                                       // slab of this 16-row tile
    if (which == 0) {
        // =================================== inverse model ===================================
        const float* P = u.params + u.inv_off;
        float* slab = u.slabs + sidx * u.total + u.inv_off;
        auto offW = [&](int l) -> long { return l == 0 ? 0 : 2L * H * H + H + (long)(l - 1) * (H * H + H); };
        auto offB = [&](int l) -> long { return offW(l) + (l == 0 ? 2L * H * H : (l < depth ? (long)H * H : (long)A * H)); };
        for (int i = tid; i < A * H; i += kThreadsU) sWout[i] = P[offW(depth) + i];
        if (tid < A) sBout[tid] = P[offB(depth) + tid];
        // layer 0 over the two K = H halves of cat(enc_1, enc_2)
        if (has_tile) {
            const int o = n0 + (lane & 15);
            const float bv = P[offB(0) + o];
            if (depth > 1) load_fwd_lines_ld<HT>(P + offW(1), H, n0, lane, sc);
            else load_dgrad_frags_buf_ld<HT>(P, 2 * H, n0, lane, sc);              // (no hidden layer: layer 0's first dgrad half)
            f32x4 acc = mfma_rows_x_lines<HT>(sE1, HS, lane, sa, bv, sScr);
            acc += mfma_rows_x_lines<HT>(sE2, HS, lane, sb, 0.f, sScr);
            fwd_tile_store<HT>(acc, sH, act, wave, lane);
        }
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l < depth; ++l) {                  // sc: this layer's set; sa: the next one
            if (has_tile) {
                const float bv = P[offB(l) + n0 + (lane & 15)];
                if (l + 1 < depth) load_fwd_lines_ld<HT>(P + offW(l + 1), H, n0, lane, sa);
                else load_dgrad_frags_buf_ld<HT>(P + offW(l), H, n0, lane, sa);   // the backward pass begins with this layer
                fwd_tile_store<HT>(mfma_rows_x_lines<HT>(sH + (long)(l - 1) * kRows * HS, HS, lane, sc, bv, sScr),
                                   sH + (long)l * kRows * HS, act, wave, lane);
#pragma unroll
                for (int c = 0; c < HT; ++c) sc[c] = sa[c];
            }
            __syncthreads();
        }
        // sc now holds the first dgrad set of the backward pass (layer depth - 1, or layer 0's first half)
        const float* Hlast = sH + (long)(depth - 1) * kRows * HS;
        if (u.split) for (int l = 0; l < depth; ++l) icm_publish<H>(sH + (long)l * kRows * HS, HS, u.hI + (long)l * u.Bpad * H, g, tid);
        // output layer (A <= 8): VALU from LDS + 16-lane reductions
        if (tid < 256) {
            const int s = tid >> 4, part = tid & 15;
            for (int k = 0; k < A; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < HT; ++i) acc = fmaf(Hlast[s * HS + part + 16 * i], sWout[k * H + part + 16 * i], acc);
                acc = group16_sum(acc);
                if (part == 0) sOut[s * kMaxOut + k] = acc + sBout[k];
            }
        }
        __syncthreads();
        // loss + d(out)
        if (wave == 0) {
            const int s = lane;
            const bool live = s < kRows && sRow[s] >= 0;
            float part = 0.f;
            if (live && u.discrete) {
                // icm.py:404-409: softmax output fed to CrossEntropyLoss (a second log-softmax)
                float q[8];
                float m = -INFINITY;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < A) m = fmaxf(m, sOut[s * kMaxOut + k]);
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) { q[k] = k < A ? expf(sOut[s * kMaxOut + k] - m) : 0.f; ssum += q[k]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) q[k] /= ssum;
                float m2 = -INFINITY;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < A) m2 = fmaxf(m2, q[k]);
                float e2[8], s2 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) { e2[k] = k < A ? expf(q[k] - m2) : 0.f; s2 += e2[k]; }
                const int a = reinterpret_cast<const int*>(sAct)[s * 8];
                float qa = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k == a) qa = q[k];
                part = -(qa - m2 - logf(s2));
                const float sc = u.icm_beta / (float)B;
                float dq[8], dot = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dq[k] = k < A ? sc * (e2[k] / s2 - (k == a ? 1.f : 0.f)) : 0.f;
                    dot += dq[k] * q[k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sDOut[s * kMaxOut + k] = q[k] * (dq[k] - dot);
            } else if (live) {
                // icm.py:411-413: mean squared error over B x A
                const float sc = u.icm_beta * 2.0f / ((float)B * (float)A);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float dv = 0.f;
                    if (k < A) {
                        const float diff = sOut[s * kMaxOut + k] - sAct[s * 8 + k];
                        part += diff * diff;
                        dv = sc * diff;
                    }
                    sDOut[s * kMaxOut + k] = dv;
                }
            } else if (s < kRows) {
#pragma unroll
                for (int k = 0; k < 8; ++k) sDOut[s * kMaxOut + k] = 0.f;
            }
            float v = lane < kRows ? part : 0.f;
            v = group16_sum(v);
            if (lane == 0) u.loss_partials[g * 2 + 0] = v;
        }
        __syncthreads();
        // output layer backward
        if (u.split) {
            if (tid < kRows * 16) {                             // d(out) rows, zero padded to 16 columns
                const int s = tid >> 4, k = tid & 15;
                u.oI[((long)g * kRows + s) * 16 + k] = k < 8 ? sDOut[s * kMaxOut + k] : 0.f;
            }
        } else if (tid < H) {
            const int i = tid;
            float h[kRows];
#pragma unroll
            for (int s = 0; s < kRows; ++s) h[s] = Hlast[s * HS + i];
            for (int k = 0; k < A; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc = fmaf(sDOut[s * kMaxOut + k], h[s], acc);
                slab[offW(depth) + (long)k * H + i] = acc;
            }
        }
        if (!u.split && tid >= 256 && tid < 256 + 8) {
            const int k = tid - 256;                           // the padded bias entries get zeros
            float acc = 0.f;
            if (k < A)
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc += sDOut[s * kMaxOut + k];
            if (k < ((A + 3) & ~3)) slab[offB(depth) + k] = acc;
        }
        if (tid >= 256) {
            const int t2 = tid - 256;
            const int s = t2 >> 4, ig = t2 & 15;
            float d[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = sDOut[s * kMaxOut + k];
            float hv[HT], acc[HT];
#pragma unroll
            for (int ii = 0; ii < HT; ++ii) { hv[ii] = Hlast[s * HS + ig + 16 * ii]; acc[ii] = 0.f; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < A) {
#pragma unroll
                    for (int ii = 0; ii < HT; ++ii) acc[ii] = fmaf(d[k], sWout[k * H + ig + 16 * ii], acc[ii]);
                }
            }
#pragma unroll
            for (int ii = 0; ii < HT; ++ii) sD0[s * HS + ig + 16 * ii] = acc[ii] * act_bwd(hv[ii], act);
        }
        __syncthreads();
        float* Dc = sD0;
        float* Dn = sD1;
#pragma unroll 1
        for (int l = depth - 1; l >= 1; --l) {             // sc: this layer's dgrad set; sa: the next one
            const float* Hin = sH + (long)(l - 1) * kRows * HS;
            if (has_tile) {
                if (l - 1 >= 1) load_dgrad_frags_buf_ld<HT>(P + offW(l - 1), H, n0, lane, sa);
                else load_dgrad_frags_buf_ld<HT>(P, 2 * H, n0, lane, sa);             // layer 0's first half
                dgrad_tile_store<HT>(mfma_rows_x_frags<HT>(Dc, HS, lane, sc, 0.f), Hin, act, Dn, nullptr, wave, lane);
#pragma unroll
                for (int c = 0; c < HT; ++c) sc[c] = sa[c];
            }
            if (u.split) icm_publish<H>(Dc, HS, u.dI + (long)l * u.Bpad * H, g, tid);
            else layer_wgrad<HT>(Dc, Hin, HS, HT, H, slab + offW(l), H, slab + offB(l), wave, lane, tid);
            __syncthreads();
            float* t = Dc; Dc = Dn; Dn = t;
        }
        // layer 0: two K halves; the gradients of the encodings leave through scratch
        if (has_tile) load_dgrad_frags_buf_ld<HT>(P + H, 2 * H, n0, lane, sa);
        if (u.split) icm_publish<H>(Dc, HS, u.dI, g, tid);
        else {
            layer_wgrad<HT>(Dc, sE1, HS, HT, H, slab + offW(0), 2 * H, slab + offB(0), wave, lane, tid);
            layer_wgrad<HT>(Dc, sE2, HS, HT, H, slab + offW(0) + H, 2 * H, nullptr, wave, lane, tid);
        }
        if (has_tile) {
            // FUSED: d enc_1 stays here (sE1: this workgroup runs the obs stream's encoder backward), d enc_2 goes to the partner (sE2)
            float* dE = u.dEnc + ((long)(0 * 2 + 0) * u.Bpad + (long)g * kRows) * H;
            dgrad_tile_store<HT>(mfma_rows_x_frags<HT>(Dc, HS, lane, sc, 0.f), nullptr, act, FUSED ? sE1 : nullptr, dE, wave, lane);
            dE = u.dEnc + ((long)(0 * 2 + 1) * u.Bpad + (long)g * kRows) * H;
            dgrad_tile_store<HT>(mfma_rows_x_frags<HT>(Dc, HS, lane, sa, 0.f), nullptr, act, FUSED ? sE2 : nullptr, dE, wave, lane);
        }
    } else {
        // =================================== forward model ===================================
        const float* P = u.params + u.fwd_off;
        float* slab = u.slabs + sidx * u.total + u.fwd_off;
        const long ld0 = H + Ain;
        auto offW = [&](int l) -> long { return l == 0 ? 0 : (long)H * ld0 + H + (long)(l - 1) * (H * H + H); };
        auto offB = [&](int l) -> long { return offW(l) + (l == 0 ? (long)H * ld0 : (long)H * H); };
        __syncthreads();                                        // sXa / sAct complete
        if (has_tile) {
            const int o = n0 + (lane & 15);
            const float bv = P[offB(0) + o];
            const float* wrow = P + (long)o * ld0 + H;
            float bq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * j + (lane >> 4);
                bq[j] = k < Ain ? wrow[k] : 0.f;
            }
            load_fwd_lines_ld<HT>(P + offW(1), H, n0, lane, sc);       // layer 1: a hidden layer, or the H -> H output layer
            f32x4 acc = mfma_rows_x_frags<HT>(sE1, HS, lane, sa, bv);
            const float* arow = sXa + (lane & 15) * kXS;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j + (lane >> 4)], bq[j], acc, 0, 0, 0);
            fwd_tile_store<HT>(acc, sH, act, wave, lane);
        }
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l < depth; ++l) {                  // sc: this layer's set; sa: the next one (l + 1 <= depth: the output layer)
            if (has_tile) {
                const float bv = P[offB(l) + n0 + (lane & 15)];
                load_fwd_lines_ld<HT>(P + offW(l + 1), H, n0, lane, sa);
                fwd_tile_store<HT>(mfma_rows_x_lines<HT>(sH + (long)(l - 1) * kRows * HS, HS, lane, sc, bv, sScr),
                                   sH + (long)l * kRows * HS, act, wave, lane);
#pragma unroll
                for (int c = 0; c < HT; ++c) sc[c] = sa[c];
            }
            __syncthreads();
        }
        const float* Hlast = sH + (long)(depth - 1) * kRows * HS;
        if (u.split) {
            for (int l = 0; l < depth; ++l) icm_publish<H>(sH + (long)l * kRows * HS, HS, u.hF + (long)l * u.Bpad * H, g, tid);
            if (tid < kRows * 16) {                             // the action columns of layer 0's input
                const int s = tid >> 4, k = tid & 15;
                u.aF[((long)g * kRows + s) * 16 + k] = sXa[s * kXS + k];
            }
        }
        // output layer H -> H (linear): the prediction of enc_2, into sD0; its dgrad set is the backward pass's first
        if (has_tile) {
            const float bv = P[offB(depth) + n0 + (lane & 15)];
            load_dgrad_frags_buf_ld<HT>(P + offW(depth), H, n0, lane, sa);
            fwd_tile_store<HT>(mfma_rows_x_lines<HT>(Hlast, HS, lane, sc, bv, sScr), sD0, -1, wave, lane);
#pragma unroll
            for (int c = 0; c < HT; ++c) sc[c] = sa[c];
        }
        __syncthreads();
        // K8: f_loss = 0.5 mean((pred - enc_2)^2); d pred = (1 - beta) (pred - enc_2) / (B H); d enc_2 = -d pred
        {
            const float sc = (1.0f - u.icm_beta) / ((float)B * (float)H);
            float part = 0.f;
            float* dE2 = u.dEnc + ((long)(1 * 2 + 1) * u.Bpad + (long)g * kRows) * H;
            for (int idx = tid; idx < kRows * H; idx += kThreadsU) {
                const int s = idx / H, i = idx - s * H;
                float dv = 0.f;
                if (sRow[s] >= 0) {
                    const float diff = sD0[s * HS + i] - sE2[s * HS + i];
                    part += diff * diff;
                    dv = sc * diff;
                }
                sD0[s * HS + i] = dv;
                if (FUSED) sE1[s * HS + i] = -dv;            // d enc_2 stays here: this workgroup runs the next_obs stream's encoder backward
                else dE2[(long)s * H + i] = -dv;
            }
            part = block_sum(part, red);
            if (tid == 0) u.loss_partials[g * 2 + 1] = 0.5f * part;
        }
        __syncthreads();
        float* Dc = sD0;
        float* Dn = sD1;
        // output layer backward, then the hidden layers
#pragma unroll 1
        for (int l = depth; l >= 1; --l) {                 // sc: this layer's dgrad set; sa: the next one
            const float* Hin = sH + (long)(l - 1) * kRows * HS;
            if (has_tile) {
                if (l - 1 >= 1) load_dgrad_frags_buf_ld<HT>(P + offW(l - 1), H, n0, lane, sa);
                else load_dgrad_frags_buf_ld<HT>(P, ld0, n0, lane, sa);               // layer 0 (the encoding columns)
                dgrad_tile_store<HT>(mfma_rows_x_frags<HT>(Dc, HS, lane, sc, 0.f), Hin, act, Dn, nullptr, wave, lane);
#pragma unroll
                for (int c = 0; c < HT; ++c) sc[c] = sa[c];
            }
            if (u.split) icm_publish<H>(Dc, HS, u.dF + (long)l * u.Bpad * H, g, tid);
            else layer_wgrad<HT>(Dc, Hin, HS, HT, H, slab + offW(l), H, slab + offB(l), wave, lane, tid);
            __syncthreads();
            float* t = Dc; Dc = Dn; Dn = t;
        }
        if (u.split) icm_publish<H>(Dc, HS, u.dF, g, tid);
        else {
            layer_wgrad<HT>(Dc, sE1, HS, HT, H, slab + offW(0), (int)ld0, slab + offB(0), wave, lane, tid);
            layer_wgrad<HT>(Dc, sXa, kXS, 1, Ain, slab + offW(0) + H, (int)ld0, nullptr, wave, lane, tid);
        }
        if (has_tile) {
            float* dE = u.dEnc + ((long)(1 * 2 + 0) * u.Bpad + (long)g * kRows) * H;
            dgrad_tile_store<HT>(mfma_rows_x_frags<HT>(Dc, HS, lane, sc, 0.f), nullptr, act, FUSED ? sE2 : nullptr, dE, wave, lane);   // FUSED: d enc_1 goes to the partner
        }
    }
    if constexpr (FUSED && H == 128) {
        // the encoding gradients change hands: sE2 holds what the partner's stream needs, fx.recv takes what mine needs;
        // sE1 (this model's gradient of my stream's encoding) stays -- the caller adds the two in the chain's order
        __syncthreads();
        pair_send(fx.block(1, g, which), sE2, HS, 0, fx.tag, tid);
        pair_recv(fx.block(1, g, which ^ 1), fx.recv, HS, 0, fx.tag, tid, reinterpret_cast<unsigned*>(fx.base));
        __syncthreads();
    }
}

template <int HT>
__global__ __launch_bounds__(kThreadsU) void icm_heads_kernel(IcmDev u) {
    constexpr int H = 16 * HT, HS = H + 4;
    const int vb = icm_block(u);
    if (vb < 0 || vb >= 2 * u.nT) return;
    float* smem = reinterpret_cast<float*>(icm_smem);
    const int dmax = u.d_inv > u.d_fwd ? u.d_inv : u.d_fwd;
    icm_heads_body<HT, false>(u, vb & 1, vb >> 1, smem, smem + 160 + 8 * H + kRows * kXS + (4L + dmax) * kRows * HS + 2 * kRows * kMaxOut, IcmFuse());
}

// ------------------------------------------------------------------------------------------------
// rollout-time intrinsic reward (ppo_policy.py:954-1007 -> icm.py:375-430 without the inverse model):
// forward model on the encodings left in scratch by icm_encoder_fwd_kernel;
// intr[row] = weight * (reward_scale / 2) * sum_d (pred - enc_2)^2.  One workgroup per 16 rows.
// ------------------------------------------------------------------------------------------------
template <int HT>
__global__ __launch_bounds__(kThreadsU) void icm_reward_kernel(IcmDev u, float scale, float* __restrict__ intr_out) {
    constexpr int H = 16 * HT, HS = H + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int act = u.act, A = u.A, Ain = u.Ain, depth = u.d_fwd;
    float* smem = reinterpret_cast<float*>(icm_smem);
    float* sXa = smem;                                        // [16, kXS]
    float* sE1 = sXa + kRows * kXS;                           // [16, HS]
    float* sE2 = sE1 + kRows * HS;
    float* sH = sE2 + kRows * HS;                             // 2 x [16, HS] ping-pong
    for (int i = tid; i < kRows * kXS; i += kThreadsU) sXa[i] = 0.f;
    for (int idx = tid; idx < 2 * kRows * (H / 4); idx += kThreadsU) {
        const int e = idx / (kRows * (H / 4)), rem = idx - e * (kRows * (H / 4));
        const int s = rem / (H / 4), c4 = rem - s * (H / 4);
        const float4 v = *reinterpret_cast<const float4*>(
            u.actE + (((long)(e * 4 + 3) * u.Bpad) + (long)g * kRows + s) * H + 4 * c4);
        *reinterpret_cast<float4*>((e == 0 ? sE1 : sE2) + s * HS + 4 * c4) = v;
    }
    __syncthreads();
    if (tid < kRows) {
        const long row = (long)g * kRows + tid;
        if (row < u.B) {
            if (u.discrete) {
                int a = (int)reinterpret_cast<const int64_t*>(u.actions)[row];
                a = a < 0 ? 0 : (a >= A ? A - 1 : a);
                sXa[tid * kXS + a] = 1.0f;
            } else {
                for (int d = 0; d < A; ++d) sXa[tid * kXS + d] = reinterpret_cast<const float*>(u.actions)[row * A + d];
            }
        }
    }
    __syncthreads();
    const float* P = u.params + u.fwd_off;
    const long ld0 = H + Ain;
    auto offW = [&](int l) -> long { return l == 0 ? 0 : (long)H * ld0 + H + (long)(l - 1) * (H * H + H); };
    auto offB = [&](int l) -> long { return offW(l) + (l == 0 ? (long)H * ld0 : (long)H * H); };
    for (int nt = wave; nt < HT; nt += kNW) {
        float4 fa[HT];
        load_fwd_frags_ld<HT, false>(P, ld0, nt * 16, lane, fa);
        const int o = nt * 16 + (lane & 15);
        f32x4 acc = mfma_rows_x_frags<HT>(sE1, HS, lane, fa, P[offB(0) + o]);
        const float* wrow = P + (long)o * ld0 + H;
        const float* arow = sXa + (lane & 15) * kXS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * j + (lane >> 4);
            const float bq = k < Ain ? wrow[k] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k], bq, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sH[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
    }
    __syncthreads();
    float* cur = sH;
    float* nxt = sH + kRows * HS;
#pragma unroll 1
    for (int l = 1; l <= depth; ++l) {                        // hidden layers, then the linear output layer
        layer_fwd<HT, true>(P + offW(l), H, P + offB(l), cur, nxt, l < depth ? act : -1, wave, lane);
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
    // row sums of (pred - enc_2)^2: 16 lanes per row
    if (tid < 256) {
        const int s = tid >> 4, part = tid & 15;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < HT; ++i) {
            const float d = cur[s * HS + part + 16 * i] - sE2[s * HS + part + 16 * i];
            acc = fmaf(d, d, acc);
        }
        acc = group16_sum(acc);
        const long row = (long)g * kRows + s;
        if (part == 0 && row < u.B) intr_out[row] = scale * acc;
    }
}

// ------------------------------------------------------------------------------------------------
// encoder backward: blockIdx.x = 2 * g + which (0: obs, 1: next_obs); slab index 2 * g + which
// ------------------------------------------------------------------------------------------------
// FUSED: the third phase of icm_fused_kernel -- the activations are still in LDS (same carve as the forward body), the
// encoding's gradient is the sum of `own` (this workgroup's model) and `recv` (the partner's), in the chain's order:
// inverse model's part + forward model's part.
template <int HT, bool FUSED>
__device__ __forceinline__ void icm_encoder_bwd_body(const IcmDev& u, const int which, const int g, float* smem, const float* own,
                                                     const float* recv) {
    constexpr int H = 16 * HT, HS = H + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vb = 2 * g + which;
    const int O = u.O, NT0 = (O + 15) >> 4, INP = 16 * NT0 + 4;
    const float* P = u.params + u.enc_off;
    float* slab = u.slabs + (long)vb * u.total + u.enc_off;
    auto encW = [&](int l) -> long { return l == 0 ? 0 : (long)H * O + H + (long)(l - 1) * (H * H + H); };
    auto encB = [&](int l) -> long { return encW(l) + (l == 0 ? (long)H * O : (long)H * H); };
    int* sRow = reinterpret_cast<int*>(smem);
    float* sX = smem + 16;                      // [16, INP]
    float* sH = sX + kRows * INP;               // 3 x [16, HS]
    float* sD0 = sH + 3L * kRows * HS;
    float* sD1 = sD0 + kRows * HS;
    const bool has_tile = wave < HT;
    float4 da[HT], db[HT];                      // dgrad sets, requested a phase ahead (layer 3's now)
    if (has_tile) load_dgrad_frags_buf_ld<HT>(P + encW(3), H, wave * 16, lane, da);
    if constexpr (FUSED) {
        const float* a = which == 0 ? own : recv;            // the inverse model's part first, as the separate launch adds them
        const float* b = which == 0 ? recv : own;
        for (int idx = tid; idx < kRows * (H / 4); idx += kThreadsU) {
            const int s2 = idx / (H / 4), c4 = idx - s2 * (H / 4);
            const float4 va = *reinterpret_cast<const float4*>(a + s2 * HS + 4 * c4);
            const float4 vb4 = *reinterpret_cast<const float4*>(b + s2 * HS + 4 * c4);
            *reinterpret_cast<float4*>(sD0 + s2 * HS + 4 * c4) = make_float4(va.x + vb4.x, va.y + vb4.y, va.z + vb4.z, va.w + vb4.w);
        }
    } else {
    icm_rows(u, g, tid, sRow);
    for (int i = tid; i < kRows * INP; i += kThreadsU) sX[i] = 0.f;
    for (int idx = tid; idx < 3 * kRows * (H / 4); idx += kThreadsU) {
        const int l = idx / (kRows * (H / 4)), rem = idx - l * (kRows * (H / 4));
        const int s = rem / (H / 4), c4 = rem - s * (H / 4);
        *reinterpret_cast<float4*>(sH + ((long)l * kRows + s) * HS + 4 * c4) = *reinterpret_cast<const float4*>(
            u.actE + (((long)(which * 4 + l) * u.Bpad) + (long)g * kRows + s) * H + 4 * c4);
    }
    for (int idx = tid; idx < kRows * (H / 4); idx += kThreadsU) {
        const int s = idx / (H / 4), c4 = idx - s * (H / 4);
        const long o0 = ((long)(0 * 2 + which) * u.Bpad + (long)g * kRows + s) * H + 4 * c4;
        const long o1 = ((long)(1 * 2 + which) * u.Bpad + (long)g * kRows + s) * H + 4 * c4;
        const float4 a = *reinterpret_cast<const float4*>(u.dEnc + o0);
        const float4 b = *reinterpret_cast<const float4*>(u.dEnc + o1);
        *reinterpret_cast<float4*>(sD0 + s * HS + 4 * c4) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    }
    __syncthreads();
    if (!FUSED && !u.split) {                   // layer 0's wgrad input (split chain: published by the forward kernel)
        const float* src = which == 0 ? u.obs : u.next_obs;
        for (int idx = tid; idx < kRows * O; idx += kThreadsU) {
            const int s = idx / O, i = idx - s * O;
            const int row = sRow[s];
            if (row >= 0) sX[s * INP + i] = src[(long)row * O + i];
        }
    }
    float* Dc = sD0;
    float* Dn = sD1;
#pragma unroll
    for (int l = 3; l >= 1; --l) {
        const float* Hin = sH + (long)(l - 1) * kRows * HS;
        if (has_tile) {
            if (l == 3) load_dgrad_frags_buf_ld<HT>(P + encW(2), H, wave * 16, lane, db);
            if (l == 2) load_dgrad_frags_buf_ld<HT>(P + encW(1), H, wave * 16, lane, da);
            const f32x4 acc = l == 2 ? mfma_rows_x_frags<HT>(Dc, HS, lane, db, 0.f) : mfma_rows_x_frags<HT>(Dc, HS, lane, da, 0.f);
            dgrad_tile_store<HT>(acc, Hin, u.act, Dn, nullptr, wave, lane);
        }
        if (u.split) icm_publish<H>(Dc, HS, u.dE + (long)(which * 4 + l) * u.Bpad * H, g, tid);
        else layer_wgrad<HT>(Dc, Hin, HS, HT, H, slab + encW(l), H, slab + encB(l), wave, lane, tid);
        __syncthreads();
        float* t = Dc; Dc = Dn; Dn = t;
    }
    if (u.split) icm_publish<H>(Dc, HS, u.dE + (long)(which * 4) * u.Bpad * H, g, tid);
    else layer_wgrad<HT>(Dc, sX, INP, NT0, O, slab + encW(0), O, slab + encB(0), wave, lane, tid);
}

template <int HT>
__global__ __launch_bounds__(kThreadsU) void icm_encoder_bwd_kernel(IcmDev u) {
    const int vb = icm_block(u);
    if (vb < 0 || vb >= 2 * u.nT) return;
    icm_encoder_bwd_body<HT, false>(u, vb & 1, vb >> 1, reinterpret_cast<float*>(icm_smem), nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// The three kernels above as ONE launch (split-wgrad chain, H = 128; args->fuse_kernels): workgroup (g, s) runs the encoder of
// stream s (obs | next_obs) for its 16 rows, swaps encodings with its partner (g, s ^ 1), runs model s (inverse | forward) --
// forward, losses, backward --, swaps the encoding gradients, and runs its stream's encoder backward with the activations
// still in LDS.  Two exchanges of data-tagged records (~2.5 k cycles each) instead of two launch boundaries and the
// reloads behind them; every panel the wgrad launch reads is stored as before; arithmetic and orders are the three
// kernels': bitwise the same gradients.  Partners sit on one XCD (workgroup b runs on XCD b % 8: slots 2 k and 2 k + 1).
// ------------------------------------------------------------------------------------------------
template <int HT>
__global__ __launch_bounds__(kThreadsU) void icm_fused_kernel(IcmDev u, unsigned char* records) {
    constexpr int H = 16 * HT, HS = H + 4;
    const int b = blockIdx.x, x = b & 7, j = b >> 3;
    int g;
    if (u.confine) {
        if ((x >> 2) != u.confine - 1) return;
        g = ((j >> 1) << 2) | (x & 3);
    } else {
        g = ((j >> 1) << 3) | x;
    }
    const int s = j & 1;
    if (g >= u.nT) return;
    float* smem = reinterpret_cast<float*>(icm_smem);
    const int INP = 16 * ((u.O + 15) >> 4) + 4;
    const int dmax = u.d_inv > u.d_fwd ? u.d_inv : u.d_fwd;
    float* regE = smem;                                                      // encoder carve: rows, inputs, 3 activation planes, dz x 2
    float* regH = regE + 16 + kRows * INP + 5L * kRows * HS;                 // the model's carve
    float* lines = regH + 160 + 8 * H + kRows * kXS + (4L + dmax) * kRows * HS + 2 * kRows * kMaxOut;
    float* encH = regE + 16 + kRows * INP;
    IcmFuse fx;
    fx.enc_own = encH + 3L * kRows * HS;                                     // the encoder's last plane
    fx.recv = encH + 4L * kRows * HS;                                        // (the backward phase's second dz plane: free until its first dgrad)
    fx.base = records;
    fx.tag = (unsigned)(u.cursor[0] + 1);
    icm_encoder_fwd_body<HT, true>(u, s, g, regE, lines);
    __syncthreads();
    icm_heads_body<HT, true>(u, s, g, regH, lines, fx);
    float* sE1 = regH + 160 + 8 * H + kRows * kXS;
    icm_encoder_bwd_body<HT, true>(u, s, g, regE, sE1, fx.recv);
}

// ------------------------------------------------------------------------------------------------
// reduce (+ Adam): the encoder region has 2 * nT slabs, the two models nT
// ------------------------------------------------------------------------------------------------
constexpr int kIcmRedThreads = 256;

__global__ __launch_bounds__(kIcmRedThreads) void icm_reduce_kernel(IcmDev u) {
    if (blockIdx.x == gridDim.x - 1) {                       // bookkeeping workgroup: loss -> totals, cursor++
        if (threadIdx.x >= 64) return;
        const int lane = threadIdx.x;
        float inv = 0.f, fl = 0.f;
        for (int g = lane; g < u.nT; g += 64) { inv += u.loss_partials[g * 2]; fl += u.loss_partials[g * 2 + 1]; }
        inv = wave_sum(inv); fl = wave_sum(fl);
        if (lane == 0) {
            const float n = (float)u.B;
            inv = u.discrete ? inv / n : inv / (n * (float)u.A);
            fl = fl / (n * (float)u.H);
            u.totals[0] += (double)((1.0f - u.icm_beta) * fl + u.icm_beta * inv);
            u.totals[1] += 1.0;
            u.cursor[0] += 1;
        }
        return;
    }
    const long n4 = u.total >> 2;
    const long idx = (long)blockIdx.x * kIcmRedThreads + threadIdx.x;
    if (idx >= n4) return;
    const float4* sl = reinterpret_cast<const float4*>(u.slabs);
    // the encoder region holds 2 * nT slabs (one per observation stream and tile), the two models nT
    const long p = idx * 4;
    const int ns = (p >= u.enc_off && p < u.enc_off + u.enc_size) ? 2 * u.nT : u.nT;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g0 = 0; g0 < ns; g0 += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = (g0 + k < ns) ? sl[(long)(g0 + k) * n4 + idx] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    reinterpret_cast<float4*>(u.grads)[idx] = acc;
    if (u.fused_adam) {
        const float step_size = u.loss_partials[2 * u.nT], bc2_sqrt = u.loss_partials[2 * u.nT + 1];
        const float gs = u.grad_scale;
        float4 pp = reinterpret_cast<float4*>(const_cast<float*>(u.params))[idx];
        float4 m = reinterpret_cast<float4*>(u.exp_avg)[idx];
        float4 v = reinterpret_cast<float4*>(u.exp_avg_sq)[idx];
#define PPOAF_ADAM1(cc)                                                          \
        {                                                                        \
            const float gi = acc.cc * gs;                                        \
            m.cc = u.beta1 * m.cc + (1.0f - u.beta1) * gi;                       \
            v.cc = u.beta2 * v.cc + (1.0f - u.beta2) * gi * gi;                  \
            pp.cc = pp.cc - step_size * (m.cc / (sqrtf(v.cc) / bc2_sqrt + u.adam_eps)); \
        }
        PPOAF_ADAM1(x) PPOAF_ADAM1(y) PPOAF_ADAM1(z) PPOAF_ADAM1(w)
#undef PPOAF_ADAM1
        reinterpret_cast<float4*>(const_cast<float*>(u.params))[idx] = pp;
        reinterpret_cast<float4*>(u.exp_avg)[idx] = m;
        reinterpret_cast<float4*>(u.exp_avg_sq)[idx] = v;
    }
}

// Split-wgrad chain: what ppoaf_icm_update_reduce launches when args->split_workspace is set.  Every weight gradient of
// the ICM is dW = dz^T x over all rows of the mini-batch (both observation streams for the encoder), formed ONCE here
// instead of per 16-row tile into 3 x B/16 slabs of the bucket (20 MB per mini-batch at C3): one 4-wave workgroup per 16
// output rows x 32 input columns (two f32 MFMA tiles sharing the dz operand, K = all rows: the waves take every fourth
// 16-row chunk and are folded in wave order), the biases as column sums of dz in the jobs of input piece 0.  With
// fused_adam the job applies Adam to its elements right away (the ICM update does not clip: ppo.py:2559-2562).  Jobs are
// dealt to XCDs in runs of the block-major list (workgroup b runs on XCD b % 8).  Last workgroup: loss -> totals, cursor.
// (p, m, v) of one element, requested at the start of the job -- beside the operand panels, not one cold round trip after
// the MFMAs -- and the Adam step on it (icm_reduce_kernel's arithmetic)
struct IcmPmv { float p, m, v; };
__device__ __forceinline__ IcmPmv icm_pmv_load(const IcmDev& u, long idx, bool ok) {
    IcmPmv r = {0.f, 0.f, 0.f};
    if (ok) { r.p = u.params[idx]; r.m = u.exp_avg[idx]; r.v = u.exp_avg_sq[idx]; }
    return r;
}
__device__ __forceinline__ void icm_adam1(const IcmDev& u, long idx, float g, const IcmPmv& s, float step_size, float bc2_sqrt) {
    const float gi = g * u.grad_scale;
    const float m = u.beta1 * s.m + (1.0f - u.beta1) * gi;
    const float v = u.beta2 * s.v + (1.0f - u.beta2) * gi * gi;
    const_cast<float*>(u.params)[idx] = s.p - step_size * (m / (sqrtf(v) / bc2_sqrt + u.adam_eps));
    u.exp_avg[idx] = m;
    u.exp_avg_sq[idx] = v;
}

__global__ __launch_bounds__(256) void icm_wgrad_kernel(IcmDev u, IcmWg w, int per_xcd) {
    __shared__ __attribute__((aligned(16))) float s_fold[6 * 256 + 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: chunk bases below stay in scalar registers
    if (b == 8 * per_xcd) {                                   // bookkeeping workgroup: loss -> totals, cursor++
        if (tid >= 64) return;
        float inv = 0.f, fl = 0.f;
        for (int g = lane; g < u.nT; g += 64) { inv += u.loss_partials[g * 2]; fl += u.loss_partials[g * 2 + 1]; }
        inv = wave_sum(inv); fl = wave_sum(fl);
        if (lane == 0) {
            const float n = (float)u.B;
            inv = u.discrete ? inv / n : inv / (n * (float)u.A);
            fl = fl / (n * (float)u.H);
            u.totals[0] += (double)((1.0f - u.icm_beta) * fl + u.icm_beta * inv);
            u.totals[1] += 1.0;
            u.cursor[0] += 1;
        }
        return;
    }
    const int job = w.xcd_job0[b & 7] + (b >> 3);
    if (job >= w.xcd_job0[(b & 7) + 1]) return;               // uniform per workgroup
    int bi = 0;
    for (int i = 1; i < w.n_blk; ++i) if (job >= w.blk[i].job0) bi = i;
    const IcmBlk& k = w.blk[bi];
    const int jj = job - k.job0, ot = jj / k.n_ip, ip = jj - ot * k.n_ip;
    // Operand addresses = uniform base (segment, 16-row chunk: scalar registers) + one 32-bit lane offset.  Lanes whose
    // output row or input column lies outside the block read a clamped (valid) address instead of a zero: they only feed
    // output elements that are never stored.
    const int oc = min(ot * 16 + (lane & 15), k.n_o - 1);
    const int ic0 = min(ip * 32 + (lane & 15), k.n_i - 1), ic1 = min(ip * 32 + 16 + (lane & 15), k.n_i - 1);
    // buffer loads: resource = the panel's base, scalar offset = segment + chunk + row quad (scalar ALU), vector offset =
    // the lane's constant byte offset -- no vector address arithmetic per load
    const unsigned dl = 4u * (unsigned)((lane >> 4) * k.ldd + oc);
    const unsigned xl0 = 4u * (unsigned)((lane >> 4) * k.ldx + ic0), xl1 = 4u * (unsigned)((lane >> 4) * k.ldx + ic1);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k.D), 0, 0xFFFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k.X), 0, 0xFFFFFFFF, 0x00020000);
    // wave 0 owns the output (C layout: column = lane & 15, rows 4 (lane >> 4) + r): its elements' optimiser state first
    const int i0 = ip * 32 + (lane & 15);
    const bool adam = u.fused_adam && wave == 0;
    IcmPmv s0[4], s1[4], sb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = ot * 16 + 4 * (lane >> 4) + r;
        const long base = k.w + (long)o * k.ldw;
        s0[r] = icm_pmv_load(u, base + i0, adam && o < k.n_o && i0 < k.n_i);
        s1[r] = icm_pmv_load(u, base + i0 + 16, adam && o < k.n_o && i0 + 16 < k.n_i);
    }
    const bool has_b = ip == 0 && k.b >= 0 && lane < 16 && ot * 16 + lane < k.n_o;
    sb = icm_pmv_load(u, k.b + ot * 16 + lane, adam && has_b);
    const int nc = u.nT, ncs = nc * k.n_seg;                  // 16-row chunks, over all segments
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    constexpr int MAXC = 8;                                   // B = 256, two streams: 32 chunks over 4 waves in one batch
    const unsigned d16 = 16u * (unsigned)k.ldd, x16 = 16u * (unsigned)k.ldx;       // bytes per 4 rows
    for (int c0 = wave; c0 < ncs; c0 += 4 * MAXC) {           // wave-uniform trip count
        float a[MAXC][4], x0[MAXC][4], x1[MAXC][4];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ci = c0 + 4 * c;
            if (ci < ncs) {                                   // wave-uniform
                const int sg = ci / nc, ch = ci - sg * nc;
                const unsigned sd = (unsigned)(4 * (sg * k.seg_d + (long)ch * 16 * k.ldd));
                const unsigned sx = (unsigned)(4 * (sg * k.seg_x + (long)ch * 16 * k.ldx));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dl, sd + j * d16, 0));
                    x0[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl0, sx + j * x16, 0));
                    x1[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl1, sx + j * x16, 0));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c0 + 4 * c < ncs) {                           // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x0[c][j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x1[c][j], acc1, 0, 0, 0);
                    bsum += a[c][j];
                }
            }
        }
    }
    if (wave > 0) {
        *reinterpret_cast<f32x4*>(s_fold + (((wave - 1) * 2 + 0) * 64 + lane) * 4) = acc0;
        *reinterpret_cast<f32x4*>(s_fold + (((wave - 1) * 2 + 1) * 64 + lane) * 4) = acc1;
    }
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 16) s_fold[1536 + wave * 16 + lane] = bsum;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int ww = 0; ww < 3; ++ww) {
        acc0 += *reinterpret_cast<const f32x4*>(s_fold + ((ww * 2 + 0) * 64 + lane) * 4);
        acc1 += *reinterpret_cast<const f32x4*>(s_fold + ((ww * 2 + 1) * 64 + lane) * 4);
    }
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (u.fused_adam) { step_size = u.loss_partials[2 * u.nT]; bc2_sqrt = u.loss_partials[2 * u.nT + 1]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = ot * 16 + 4 * (lane >> 4) + r;
        if (o < k.n_o) {
            const long base = k.w + (long)o * k.ldw;
            if (i0 < k.n_i) { u.grads[base + i0] = acc0[r]; if (u.fused_adam) icm_adam1(u, base + i0, acc0[r], s0[r], step_size, bc2_sqrt); }
            if (i0 + 16 < k.n_i) { u.grads[base + i0 + 16] = acc1[r]; if (u.fused_adam) icm_adam1(u, base + i0 + 16, acc1[r], s1[r], step_size, bc2_sqrt); }
        }
    }
    if (has_b) {
        const float bg = s_fold[1536 + lane] + s_fold[1536 + 16 + lane] + s_fold[1536 + 32 + lane] + s_fold[1536 + 48 + lane];
        const long idx = k.b + ot * 16 + lane;
        u.grads[idx] = bg;
        if (u.fused_adam) icm_adam1(u, idx, bg, sb, step_size, bc2_sqrt);
    }
}

static int jobs_of(const IcmBlk* blk, int i, int n, int jobs) { return (i + 1 < n ? blk[i + 1].job0 : jobs) - blk[i].job0; }
// panel layout of the split workspace and the block table of the wgrad launch -> bytes
static size_t icm_split_layout(IcmDev& u, char* base, IcmWg* w) {
    const long H = u.H, plane = u.Bpad * H;
    size_t off = 0;
    auto take = [&](size_t floats) { float* p = reinterpret_cast<float*>(base + off); off += (floats * 4 + 255) & ~(size_t)255; return p; };
    u.XO = 16 * ((u.O + 15) / 16);
    u.xO = take((size_t)2 * u.Bpad * u.XO);
    u.dE = take((size_t)8 * plane);
    u.hI = take((size_t)u.d_inv * plane); u.dI = take((size_t)u.d_inv * plane); u.oI = take((size_t)u.Bpad * 16);
    u.hF = take((size_t)u.d_fwd * plane); u.dF = take((size_t)(u.d_fwd + 1) * plane); u.aF = take((size_t)u.Bpad * 16);
    if (!w) return off;
    int n = 0, jobs = 0;
    auto add = [&](const float* D, int ldd, long seg_d, const float* X, int ldx, long seg_x, int n_seg, int n_o, int n_i,
                   long wo, int ldw, long bo) {
        IcmBlk& k = w->blk[n++];
        k.D = D; k.X = X; k.seg_d = seg_d; k.seg_x = seg_x; k.w = wo; k.b = bo;
        k.n_seg = n_seg; k.ldd = ldd; k.ldx = ldx; k.n_o = n_o; k.n_i = n_i; k.ldw = ldw; k.job0 = jobs;
        k.n_ip = (n_i + 31) / 32;
        jobs += ((n_o + 15) / 16) * k.n_ip;
    };
    const int Hi = (int)H;
    // encoder (both observation streams: two segments): layer 0 from the gathered rows, layers 1..3 from the scratch
    long e = u.enc_off;
    add(u.dE, Hi, 4 * plane, u.xO, u.XO, u.Bpad * u.XO, 2, Hi, u.O, e, u.O, e + H * u.O);
    e += H * u.O + H;
    for (int l = 1; l < 4; ++l) {
        add(u.dE + l * plane, Hi, 4 * plane, u.actE + (l - 1) * plane, Hi, 4 * plane, 2, Hi, Hi, e, Hi, e + H * H);
        e += H * H + H;
    }
    // inverse model: layer 0 over cat(enc_1, enc_2), hidden layers, output layer (A rows)
    long q = u.inv_off;
    add(u.dI, Hi, 0, u.actE + 3 * plane, Hi, 0, 1, Hi, Hi, q, 2 * Hi, q + 2 * H * H);
    add(u.dI, Hi, 0, u.actE + 7 * plane, Hi, 0, 1, Hi, Hi, q + H, 2 * Hi, -1);
    q += 2 * H * H + H;
    for (int l = 1; l < u.d_inv; ++l) {
        add(u.dI + l * plane, Hi, 0, u.hI + (l - 1) * plane, Hi, 0, 1, Hi, Hi, q, Hi, q + H * H);
        q += H * H + H;
    }
    add(u.oI, 16, 0, u.hI + (u.d_inv - 1) * plane, Hi, 0, 1, u.A, Hi, q, Hi, q + (long)u.A * H);
    // forward model: layer 0 over cat(enc_1, action), hidden layers, output layer (H rows)
    long f = u.fwd_off;
    const int ld0 = Hi + u.Ain;
    add(u.dF, Hi, 0, u.actE + 3 * plane, Hi, 0, 1, Hi, Hi, f, ld0, f + H * ld0);
    add(u.dF, Hi, 0, u.aF, 16, 0, 1, Hi, u.Ain, f + H, ld0, -1);
    f += H * ld0 + H;
    for (int l = 1; l <= u.d_fwd; ++l) {
        add(u.dF + l * plane, Hi, 0, u.hF + (l - 1) * plane, Hi, 0, 1, Hi, Hi, f, Hi, f + H * H);
        f += H * H + H;
    }
    w->n_blk = n; w->n_jobs = jobs;
    // deal the block-major job list to the 8 XCDs in runs of equal COST (a job's cost = its K: the encoder's jobs run over
    // both observation streams), so that no XCD is left with twice the work of the others
    long total = 0;
    for (int i = 0; i < n; ++i) total += (long)(jobs_of(w->blk, i, n, jobs)) * w->blk[i].n_seg;
    long acc = 0;
    int x = 0;
    w->xcd_job0[0] = 0;
    for (int i = 0; i < n; ++i) {
        const int nj = jobs_of(w->blk, i, n, jobs);
        for (int j = 0; j < nj; ++j) {
            while (x < 7 && acc * 8 >= total * (x + 1)) w->xcd_job0[++x] = w->blk[i].job0 + j;
            acc += w->blk[i].n_seg;
        }
    }
    while (x < 8) w->xcd_job0[++x] = jobs;
    return off;
}

// args->fuse_kernels: the exchange records of icm_fused_kernel come FIRST in the split workspace (a fixed place, whatever the
// mini-batch size), the panels behind them
constexpr size_t kIcmRecBytes = kPairHeaderBytes + 2 * (size_t)kPairMaxTiles * 2 * kPairRecBytes;
static_assert(kIcmRecBytes % 256 == 0, "panels stay 256-byte aligned");
static char* icm_panels_base(const ppoaf_icm_update_args_t* a) {
    return reinterpret_cast<char*>(a->split_workspace) + (a->fuse_kernels ? kIcmRecBytes : 0);
}

static int make_icm(const ppoaf_icm_update_args_t* a, IcmDev& u, bool training = true) {
    PPOAF_REQUIRE(a, "icm_update: null args");
    PPOAF_REQUIRE(a->hidden == 64 || a->hidden == 128, "icm_update: hidden=%d is not an instantiated width (64, 128)", a->hidden);
    PPOAF_REQUIRE(a->obs_dim >= 1 && a->obs_dim <= 1024, "icm_update: obs_dim=%d", a->obs_dim);
    PPOAF_REQUIRE(a->action_dim >= 1 && a->action_dim <= 8 && a->fwd_action_dim >= 1 && a->fwd_action_dim <= 8,
                  "icm_update: action_dim=%d fwd_action_dim=%d must be in [1,8]", a->action_dim, a->fwd_action_dim);
    PPOAF_REQUIRE(a->depth_inv >= 1 && a->depth_inv <= 3 && a->depth_fwd >= 1 && a->depth_fwd <= 3,
                  "icm_update: hidden depths (%d, %d) out of [1,3]", a->depth_inv, a->depth_fwd);
    PPOAF_REQUIRE(a->activation >= 0 && a->activation <= 2, "icm_update: activation=%d", a->activation);
    PPOAF_REQUIRE(a->B >= 1 && a->batch_stride >= a->B, "icm_update: B=%ld stride=%ld", (long)a->B, (long)a->batch_stride);
    PPOAF_REQUIRE(a->params && a->obs && a->next_obs && a->actions && a->act_scratch, "icm_update: null pointer");
    PPOAF_REQUIRE(((uintptr_t)a->params & 15) == 0 && ((uintptr_t)a->act_scratch & 15) == 0,
                  "icm_update: buckets and scratch must be 16-byte aligned");
    if (training) {
        PPOAF_REQUIRE(a->grads && a->exp_avg && a->exp_avg_sq && a->slabs && a->step_count && a->lr &&
                          (a->perm || a->inputs_in_batch_order) && a->cursor && a->denc_scratch && a->loss_partials && a->totals,
                      "icm_update: null pointer");
        PPOAF_REQUIRE(((uintptr_t)a->grads & 15) == 0 && ((uintptr_t)a->slabs & 15) == 0 &&
                          ((uintptr_t)a->exp_avg & 15) == 0 && ((uintptr_t)a->exp_avg_sq & 15) == 0 &&
                          ((uintptr_t)a->denc_scratch & 15) == 0,
                      "icm_update: buckets and scratch must be 16-byte aligned");
    }
    const long H = a->hidden, O = a->obs_dim, A = a->action_dim, Ain = a->fwd_action_dim;
    auto pad4 = [](long x) { return (x + 3) / 4 * 4; };
    const long enc_size = H * O + H + 3 * (H * H + H);
    const long inv_size = 2 * H * H + H + (long)(a->depth_inv - 1) * (H * H + H) + A * H + pad4(A);
    const long fwd_size = H * (H + Ain) + H + (long)(a->depth_fwd - 1) * (H * H + H) + H * H + H;
    PPOAF_REQUIRE(a->enc_offset % 4 == 0 && a->inv_offset == a->enc_offset + enc_size &&
                      a->fwd_offset == a->inv_offset + inv_size && a->bucket_total == a->fwd_offset + fwd_size,
                  "icm_update: bucket layout (enc %ld, inv %ld, fwd %ld, total %ld) does not match the topology "
                  "(sizes %ld, %ld, %ld)", (long)a->enc_offset, (long)a->inv_offset, (long)a->fwd_offset,
                  (long)a->bucket_total, enc_size, inv_size, fwd_size);
    PPOAF_REQUIRE(a->bucket_total % 4 == 0, "icm_update: bucket_total must be a multiple of 4");
    u.O = a->obs_dim; u.H = a->hidden; u.A = a->action_dim; u.Ain = a->fwd_action_dim;
    u.d_inv = a->depth_inv; u.d_fwd = a->depth_fwd; u.act = a->activation; u.discrete = a->discrete != 0;
    u.enc_off = a->enc_offset; u.inv_off = a->inv_offset; u.fwd_off = a->fwd_offset; u.enc_size = enc_size;
    u.total = a->bucket_total;
    u.params = a->params; u.grads = a->grads; u.exp_avg = a->exp_avg; u.exp_avg_sq = a->exp_avg_sq; u.slabs = a->slabs;
    u.step_count = a->step_count; u.lr = a->lr; u.beta1 = a->beta1; u.beta2 = a->beta2; u.adam_eps = a->adam_eps;
    u.grad_scale = a->grad_scale; u.obs = a->obs; u.next_obs = a->next_obs; u.actions = a->actions;
    u.perm = a->perm; u.row_map = a->row_map; u.n_rows = a->n_rows; u.cursor = a->cursor; u.B = a->B;
    u.batch_stride = a->batch_stride;
    u.nT = (int)((a->B + kRows - 1) / kRows);
    u.Bpad = (long)u.nT * kRows;
    u.icm_beta = a->icm_beta; u.fused_adam = a->fused_adam != 0; u.pregathered = a->inputs_in_batch_order != 0;
    u.actE = a->act_scratch; u.dEnc = a->denc_scratch; u.loss_partials = a->loss_partials; u.totals = a->totals;
    PPOAF_REQUIRE(a->xcd_half >= 0 && a->xcd_half <= 2, "icm_update: xcd_half=%d (0, 1 or 2)", a->xcd_half);
    u.confine = training ? a->xcd_half : 0;
    u.split = 0; u.XO = 0;
    u.xO = u.dE = u.hI = u.dI = u.oI = u.hF = u.dF = u.aF = nullptr;
    if (training && a->split_workspace) {
        PPOAF_REQUIRE((((uintptr_t)a->split_workspace) & 255) == 0, "icm_update: split_workspace must be 256-byte aligned");
        PPOAF_REQUIRE(a->fuse_kernels == 0 || a->fuse_kernels == 1, "icm_update: fuse_kernels=%d (0 or 1)", a->fuse_kernels);
        const size_t need = icm_split_layout(u, icm_panels_base(a), nullptr) + (a->fuse_kernels ? kIcmRecBytes : 0);
        PPOAF_REQUIRE((size_t)a->split_workspace_bytes >= need, "icm_update: split_workspace of %ld B, %zu needed",
                      (long)a->split_workspace_bytes, need);
        u.split = 1;
    }
    return PPOAF_OK;
}

// gfx950 has 160 KB of LDS per CU; launches above the 64 KB default need the attribute set once per kernel
static int allow_large_lds(const void* kernel, size_t bytes, bool& done, const char* what) {
    if (bytes <= 64 * 1024 || done) return PPOAF_OK;
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, kernel);          // static __shared__ counts against the same 160 KB
    if (e != hipSuccess) { set_error("%s: hipFuncGetAttributes: %s", what, hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
    const size_t room = 160 * 1024 - fa.sharedSizeBytes;
    PPOAF_REQUIRE(bytes <= room, "%s: needs %zu B of dynamic LDS, %zu available beside %zu B of static LDS", what, bytes,
                  room, (size_t)fa.sharedSizeBytes);
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)room);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
    done = true;
    return PPOAF_OK;
}

template <int HT>
static int launch_icm_fwd_bwd(const IcmDev& u, hipStream_t s) {
    const size_t HS = 16 * HT + 4, H = 16 * HT;
    const size_t INP = 16 * ((u.O + 15) / 16) + 4;
    const size_t lds_lines = (size_t)kNW * 2 * kLineSlot * 4;                  // every wave's two line slots
    const size_t lds_enc_f = (16 + kRows * INP + 4 * kRows * HS) * 4 + lds_lines;
    const size_t lds_enc_b = (16 + kRows * INP + 5 * kRows * HS) * 4;
    const int dmax = u.d_inv > u.d_fwd ? u.d_inv : u.d_fwd;
    const size_t lds_heads = (160 + 8 * H + kRows * kXS + (4 + dmax) * kRows * HS + 2 * kRows * kMaxOut) * 4 + lds_lines;
    PPOAF_REQUIRE(lds_enc_b <= 160 * 1024 && lds_heads <= 160 * 1024 && lds_enc_f <= 160 * 1024,
                  "icm_update: needs %zu / %zu B of LDS (> 160 KiB): obs_dim or depth too large",
                  lds_enc_b, lds_heads);
    static bool big_f = false, big_h = false, big_b = false;
    int rc = allow_large_lds(reinterpret_cast<const void*>(icm_encoder_fwd_kernel<HT>), lds_enc_f, big_f, "icm_encoder_fwd");
    if (!rc) rc = allow_large_lds(reinterpret_cast<const void*>(icm_heads_kernel<HT>), lds_heads, big_h, "icm_heads");
    if (!rc) rc = allow_large_lds(reinterpret_cast<const void*>(icm_encoder_bwd_kernel<HT>), lds_enc_b, big_b, "icm_encoder_bwd");
    if (rc) return rc;
    const unsigned grid = u.confine ? 8u * (unsigned)((2 * u.nT + 3) / 4) : 2u * (unsigned)u.nT;
    hipLaunchKernelGGL(icm_encoder_fwd_kernel<HT>, dim3(grid), dim3(kThreadsU), lds_enc_f, s, u);
    rc = check_launch("icm_encoder_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(icm_heads_kernel<HT>, dim3(grid), dim3(kThreadsU), lds_heads, s, u);
    rc = check_launch("icm_heads");
    if (rc) return rc;
    hipLaunchKernelGGL(icm_encoder_bwd_kernel<HT>, dim3(grid), dim3(kThreadsU), lds_enc_b, s, u);
    return check_launch("icm_encoder_bwd");
}

}  // namespace ppoaf

using namespace ppoaf;

// 1 when these arguments run the three kernels as one launch (args->fuse_kernels, the split-wgrad chain, H = 128, LDS room)
static size_t icm_fused_lds(const IcmDev& u) {
    const size_t H = (size_t)u.H, HS = H + 4, INP = 16 * ((u.O + 15) / 16) + 4;
    const int dmax = u.d_inv > u.d_fwd ? u.d_inv : u.d_fwd;
    return (16 + kRows * INP + 5 * kRows * HS + 160 + 8 * H + kRows * kXS + (4 + dmax) * kRows * HS + 2 * kRows * kMaxOut +
            (size_t)kNW * 2 * kLineSlot) * 4;
}
static bool icm_fuses(const ppoaf_icm_update_args_t* a, const IcmDev& u) {
    return a->fuse_kernels && u.split && u.H == 128 && u.nT <= kPairMaxTiles && icm_fused_lds(u) + 256 <= 160 * 1024;
}
extern "C" int ppoaf_icm_update_fuses_kernels(const ppoaf_icm_update_args_t* args) {
    IcmDev u;
    if (make_icm(args, u)) return -1;
    return icm_fuses(args, u) ? 1 : 0;
}

extern "C" int ppoaf_icm_update_fwd_bwd(const ppoaf_icm_update_args_t* args, ppoaf_stream_t stream) {
    IcmDev u;
    const int rc = make_icm(args, u);
    if (rc) return rc;
    if (icm_fuses(args, u)) {
        const size_t lds = icm_fused_lds(u);
        static bool big = false;
        int rc2 = allow_large_lds(reinterpret_cast<const void*>(icm_fused_kernel<8>), lds, big, "icm_fused");
        if (rc2) return rc2;
        const unsigned slots = 2u * (unsigned)(u.confine ? (u.nT + 3) / 4 : (u.nT + 7) / 8);
        hipLaunchKernelGGL(icm_fused_kernel<8>, dim3(8u * slots), dim3(kThreadsU), lds, (hipStream_t)stream, u,
                           reinterpret_cast<unsigned char*>(args->split_workspace));
        return check_launch("icm_fused");
    }
    if (u.H == 64) return launch_icm_fwd_bwd<4>(u, (hipStream_t)stream);
    return launch_icm_fwd_bwd<8>(u, (hipStream_t)stream);
}

extern "C" int ppoaf_icm_update_reduce(const ppoaf_icm_update_args_t* args, ppoaf_stream_t stream) {
    IcmDev u;
    const int rc = make_icm(args, u);
    if (rc) return rc;
    if (u.split) {
        IcmWg w;
        icm_split_layout(u, icm_panels_base(args), &w);
        int per_xcd = 1;
        for (int x = 0; x < 8; ++x) per_xcd = std::max(per_xcd, w.xcd_job0[x + 1] - w.xcd_job0[x]);
        hipLaunchKernelGGL(icm_wgrad_kernel, dim3((unsigned)(8 * per_xcd + 1)), dim3(256), 0, (hipStream_t)stream, u, w, per_xcd);
        return check_launch("icm_update_reduce (wgrad)");
    }
    const long n4 = u.total >> 2;
    hipLaunchKernelGGL(icm_reduce_kernel, dim3((unsigned)((n4 + kIcmRedThreads - 1) / kIcmRedThreads) + 1u),
                       dim3(kIcmRedThreads), 0, (hipStream_t)stream, u);
    return check_launch("icm_reduce");
}

extern "C" int ppoaf_icm_update_split_workspace_bytes(const ppoaf_icm_update_args_t* args, int64_t* bytes_out) {
    PPOAF_REQUIRE(args && bytes_out, "icm_update_split_workspace_bytes: null argument");
    ppoaf_icm_update_args_t a = *args;
    a.split_workspace = nullptr; a.split_workspace_bytes = 0;
    IcmDev u;
    const int rc = make_icm(&a, u);
    if (rc) return rc;
    *bytes_out = (int64_t)(icm_split_layout(u, nullptr, nullptr) + (args->fuse_kernels ? kIcmRecBytes : 0));
    return PPOAF_OK;
}

template <int HT>
static int launch_icm_reward(const IcmDev& u, float scale, float* intr_out, hipStream_t s) {
    const size_t HS = 16 * HT + 4;
    const size_t INP = 16 * ((u.O + 15) / 16) + 4;
    const size_t lds_enc_f = (16 + kRows * INP + 4 * kRows * HS) * 4 + (size_t)kNW * 2 * kLineSlot * 4;
    const size_t lds_rew = (kRows * kXS + 4 * kRows * HS) * 4;
    PPOAF_REQUIRE(lds_enc_f <= 160 * 1024 && lds_rew <= 160 * 1024, "icm_intrinsic_reward: obs_dim too large");
    static bool big_f = false, big_r = false;
    int rc = allow_large_lds(reinterpret_cast<const void*>(icm_encoder_fwd_kernel<HT>), lds_enc_f, big_f, "icm_encoder_fwd");
    if (!rc) rc = allow_large_lds(reinterpret_cast<const void*>(icm_reward_kernel<HT>), lds_rew, big_r, "icm_reward");
    if (rc) return rc;
    hipLaunchKernelGGL(icm_encoder_fwd_kernel<HT>, dim3(2u * (unsigned)u.nT), dim3(kThreadsU), lds_enc_f, s, u);
    rc = check_launch("icm_intrinsic_reward/encoder");
    if (rc) return rc;
    hipLaunchKernelGGL(icm_reward_kernel<HT>, dim3((unsigned)u.nT), dim3(kThreadsU), lds_rew, s, u, scale, intr_out);
    return check_launch("icm_intrinsic_reward");
}

extern "C" int ppoaf_icm_intrinsic_reward(const ppoaf_icm_update_args_t* args, float scale, float* intr_out,
                                          ppoaf_stream_t stream) {
    IcmDev u;
    const int rc = make_icm(args, u, false);
    if (rc) return rc;
    PPOAF_REQUIRE(intr_out, "icm_intrinsic_reward: null output");
    PPOAF_REQUIRE(args->perm == nullptr && args->fused_adam == 0,
                  "icm_intrinsic_reward: rows are the batch itself (perm must be NULL, fused_adam 0)");
    if (u.H == 64) return launch_icm_reward<4>(u, scale, intr_out, (hipStream_t)stream);
    return launch_icm_reward<8>(u, scale, intr_out, (hipStream_t)stream);
}

// mpi_avg_gradients at the ICM update's per-mini-batch call site (ppo.py:2556-2558) when the K17 peer exchange is not
// available: fwd_bwd (3 launches) -> reduce -> RCCL sum all-reduce of the gradient bucket -> K11 Adam (no clipping in
// the ICM update, ppo.py:2559-2562), for n consecutive mini-batches issued from this one call -- the host cost per
// mini-batch stays below the GPU's, as with ppoaf_ppo_update_chain_allreduce.
extern "C" int ppoaf_icm_update_chain_allreduce(const ppoaf_icm_update_args_t* args, ppoaf_comm_t* comm, int64_t n_minibatches,
                                                double* norm_scratch, float* grad_norm_out, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(args && comm && norm_scratch, "icm_update_chain_allreduce: null argument");
    PPOAF_REQUIRE(n_minibatches >= 1 && n_minibatches <= (1 << 20), "icm_update_chain_allreduce: n_minibatches=%ld", (long)n_minibatches);
    PPOAF_REQUIRE(args->fused_adam == 0, "icm_update_chain_allreduce: fused_adam must be 0 (Adam follows the all-reduce)");
    for (int64_t j = 0; j < n_minibatches; ++j) {
        int rc = ppoaf_icm_update_fwd_bwd(args, stream);
        if (rc == PPOAF_OK) rc = ppoaf_icm_update_reduce(args, stream);
        if (rc == PPOAF_OK) rc = ppoaf_allreduce_sum_f32(comm, args->grads, args->bucket_total, stream);
        if (rc == PPOAF_OK)
            rc = ppoaf_clip_adam_step(const_cast<float*>(args->params), args->grads, args->exp_avg, args->exp_avg_sq,
                                      args->bucket_total, args->step_count, args->lr, args->beta1, args->beta2, args->adam_eps,
                                      args->grad_scale, 0.0f, norm_scratch, grad_norm_out, stream);
        if (rc != PPOAF_OK) return rc;
    }
    return PPOAF_OK;
}
